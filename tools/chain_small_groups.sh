lib=avisynth_sangnom2_amd/libsangnom_hip.so
cp $lib /tmp/sn_keep.so; cp ab/tight_hooks.so $lib
for fmt in YUV420P8 YUV420P16 YUV420PS; do
for g in 1 2 4 8; do
  printf "%s G=%s: " $fmt $g
  SN_CHAIN_GROUPS=$g timeout -k 10 120 python3 tools/pool_bench.py --fmt $fmt --w 720 --h 480 --frames 1 2 3 4 6 8 16 32 --iters 50 2>/dev/null | python3 -c "
import sys, json
print(' '.join('%d:%.3f' % (json.loads(l)['frames_per_launch'], json.loads(l)['ms_per_launch']) for l in sys.stdin))"
done; done
cp /tmp/sn_keep.so $lib
