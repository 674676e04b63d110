#!/bin/bash
# like tools/ab_workloads.sh, without the verification (diagnostic builds give wrong pictures)
libs=$1; wls=$2
lib=avisynth_sangnom2_amd/libsangnom_hip.so
cp "$lib" /tmp/sn_keep.so
for wl in $wls; do
  for i in 1 2; do
    for v in $libs; do
        cp "$v" "$lib"
        printf "%s %s " "$wl" "$v"
        python3 bench.py --no-cpu-baseline --no-verify --steps 6 --warmup 2 --workload $wl 2>/dev/null | tail -1 |
            python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['frames_per_s'], d['roofline']['frac'])"
    done
  done
done
cp /tmp/sn_keep.so "$lib"
