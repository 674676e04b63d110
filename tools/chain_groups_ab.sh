#!/bin/bash
# tools/chain_groups_ab.sh <hooks-build.so> [workload]: the 480p history-carrying stream with 1, 2, 4, 8 workgroups per buffer
# (a build with -DSN_TEST_HOOKS reads SN_CHAIN_GROUPS)
lib=avisynth_sangnom2_amd/libsangnom_hip.so
cp $lib /tmp/sn_keep.so
cp "$1" $lib
wl=${2:-480p-YUV420P8}
for i in 1 2; do
  for g in 1 2 4 8; do
    printf "%s groups=%s " $wl $g
    SN_CHAIN_GROUPS=$g timeout -k 10 120 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --workload $wl 2>/dev/null | tail -1 |
        python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['frames_per_s'])"
  done
done
cp /tmp/sn_keep.so $lib
