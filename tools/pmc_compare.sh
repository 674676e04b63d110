#!/bin/bash
# tools/pmc_compare.sh <tag> <workload> [workload ...]: wider SQ counter sets for the dominant kernel of each workload, one
# rocprofv3 --pmc pass per set (never together with a trace), summarised per dispatch under gpurun_out/pmc_<tag>/.
set -e -o pipefail
tag=$1; shift
out=gpurun_out/pmc_$tag
mkdir -p "$out"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
sets=("SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
      "SQ_WAVE_CYCLES SQ_IFETCH SQ_IFETCH_LEVEL SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_ACTIVE_INST_SCA"
      "SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_IDX_ACTIVE"
      "SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES")
for wl in "$@"; do
    i=0
    for set in "${sets[@]}"; do
        i=$((i + 1))
        rocprofv3 --pmc $set --output-format csv -d "$out/${wl}_$i" -- python3 bench.py --workload $wl --steps 3 --warmup 1 --no-cpu-baseline --no-verify > "$out/${wl}_$i.log" 2>&1
    done
    python3 tools/pmc_summary.py "$out"/${wl}_[0-9] > "$out/summary_$wl.csv"
    rm -rf "$out"/${wl}_[0-9]/*/
    echo "$wl done"
done
