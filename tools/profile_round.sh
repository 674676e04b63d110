#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box, from the repo root):
#   tools/profile_round.sh r1            -> gpurun_out/prof_r1/{bench.json,kernel_stats.csv,pmc_summary.csv}
# Counters go in their own passes, never together with a trace (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -e -o pipefail
tag=${1:-r1}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
args="--steps 5 --warmup 2 --no-cpu-baseline"
python3 bench.py > "$out/bench.json" 2> "$out/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py $args > "$out/stats.log" 2>&1
find "$out/stats" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats.csv" \;
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i + 1))
    rocprofv3 --pmc $set --output-format csv -d "$out/pmc$i" -- python3 bench.py $args > "$out/pmc$i.log" 2>&1
done
python3 tools/pmc_summary.py "$out"/pmc* > "$out/pmc_summary.csv"
echo "done: $out"
