#!/bin/bash
# A/B of library builds over several bench.py workloads on ONE box:
#   tools/ab_workloads.sh "ab/a.so ab/b.so" "2160p-Y16 2160p-Y32 ..."
set -e
libs=$1; wls=$2
lib=avisynth_sangnom2_amd/libsangnom_hip.so
cp "$lib" /tmp/sn_keep.so
for wl in $wls; do
  for i in 1 2; do
    for v in $libs; do
        cp "$v" "$lib"
        printf "%s %s " "$wl" "$v"
        python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --workload $wl 2>/dev/null | tail -1 |
            python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['frames_per_s'], d['roofline']['frac'])"
    done
  done
done
cp /tmp/sn_keep.so "$lib"
