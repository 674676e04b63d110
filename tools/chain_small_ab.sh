#!/bin/bash
# tools/chain_small_ab.sh "a.so b.so": the history-carrying 720x480 clip in launches of 1 .. 64 frames with each build
lib=avisynth_sangnom2_amd/libsangnom_hip.so
cp $lib /tmp/sn_keep.so
for fmt in YUV420P8 YUV420P16; do
  for v in $1; do
    cp $v $lib
    echo "== $fmt $v"
    timeout -k 10 120 python3 tools/pool_bench.py --fmt $fmt --w 720 --h 480 --frames 1 2 4 8 16 32 64 --iters 50 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['frames_per_launch'], d['ms_per_launch'], d['frames_per_s'])"
  done
done
cp /tmp/sn_keep.so $lib
