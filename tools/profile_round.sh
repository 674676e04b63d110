#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box, from the repo root):
#   tools/profile_round.sh r2 [workload ...]     (default workload: 2160p-Y8, bench.py's headline)
# -> gpurun_out/prof_<tag>/: bench_<workload>.json (un-profiled bench line), kernel_stats_<workload>.csv
#    (rocprofv3 --kernel-trace --stats), pmc_summary_<workload>.csv (per dispatch) and counters.json (per launch,
#    what bench.py's `traffic` / `valu` fields replay).  Counters go in their own passes, never together with a
#    trace, FETCH_SIZE and WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM / rocprofv3 section).
set -e -o pipefail
tag=${1:-r2}
shift || true
wls=${@:-2160p-Y8}
out=gpurun_out/prof_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
steps=5; warm=2
for wl in $wls; do
    args="--workload $wl --steps $steps --warmup $warm --no-cpu-baseline"
    python3 bench.py --workload $wl --no-cpu-baseline > "$out/bench_$wl.json" 2> "$out/bench_$wl.err"
    frames=$(python3 -c "import json,sys; print(json.loads(open('$out/bench_$wl.json').read().strip().splitlines()[-1])['config']['frames_per_step_per_gpu'])")
    rm -rf "$out/stats"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$out/stats" -- python3 bench.py $args > "$out/stats_$wl.log" 2>&1
    find "$out/stats" -name '*kernel_stats.csv' -exec cp {} "$out/kernel_stats_$wl.csv" \;
    rm -rf "$out/stats" "$out"/pmc_${wl}_*
    i=0
    for set in "FETCH_SIZE" "WRITE_SIZE" \
               "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU"; do
        i=$((i + 1))
        rocprofv3 --pmc $set --output-format csv -d "$out/pmc_${wl}_$i" -- python3 bench.py $args > "$out/pmc_${wl}_$i.log" 2>&1
    done
    if [ "$wl" = "2160p-Y8" ]; then
        rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
            --output-format csv -d "$out/pmc_${wl}_4" -- python3 bench.py $args > "$out/pmc_${wl}_4.log" 2>&1
    fi
    python3 tools/pmc_summary.py "$out"/pmc_${wl}_* > "$out/pmc_summary_$wl.csv"
    python3 tools/pmc_summary.py --json "$out/counters.json" --workload $wl --frames $frames --steps $((steps + warm)) "$out"/pmc_${wl}_*
    rm -rf "$out"/pmc_${wl}_*/*/  # the raw per-dispatch dumps are large; the summaries stay
    echo "$wl done"
done
echo "done: $out"
