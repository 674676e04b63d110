#!/bin/bash
# A/B builds of libsangnom_hip.so on the SAME box (boxes of the pool differ by several per cent):
#   tools/ab_bench.sh "ab/prev.so ab/new.so ..." [bench.py arguments...]
# alternates the libraries three times and prints frames/s of each run.
set -e
libs=$1; shift
lib=avisynth_sangnom2_amd/libsangnom_hip.so
cp "$lib" /tmp/sn_keep.so
for i in 1 2 3; do
    for v in $libs; do
        cp "$v" "$lib"
        printf "%s " "$v"
        python3 bench.py --no-cpu-baseline --steps 8 --warmup 2 "$@" 2>/dev/null | tail -1 |
            python3 -c "import sys, json; d = json.loads(sys.stdin.read()); print(d['config']['frame'], d['frames_per_s'])"
    done
done
cp /tmp/sn_keep.so "$lib"
