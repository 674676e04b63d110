#!/bin/bash
# tools/chain_prof.sh <lib.so> <tag>: kernel durations and SQ counters of the 480p history-carrying stream with one build
set -e
lib=avisynth_sangnom2_amd/libsangnom_hip.so
[ "$1" -ef $lib ] || cp "$1" $lib
out=gpurun_out/chain_$2
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
args="--workload 480p-YUV420P8 --steps 5 --warmup 2 --no-cpu-baseline"
rm -rf $out/stats $out/pmc
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python3 bench.py $args > $out/stats.log 2>&1
find $out/stats -name '*kernel_stats.csv' -exec cp {} $out/kernel_stats.csv \;
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $out/pmc -- python3 bench.py $args > $out/pmc.log 2>&1
python3 tools/pmc_summary.py $out/pmc > $out/pmc_summary.csv
rm -rf $out/stats $out/pmc
head -4 $out/kernel_stats.csv | cut -c1-160
grep -i "chain" $out/pmc_summary.csv | cut -c1-300
