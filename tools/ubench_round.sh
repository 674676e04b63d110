#!/bin/bash
# Round-3 microbenchmark evidence (run on the GPU box from the repo root): wall-clock table + PMC cross-check.
#   tools/ubench_round.sh r3   -> gpurun_out/ubench_<tag>/{table.txt,pmc.txt}
set -e -o pipefail
tag=${1:-r3}
out=gpurun_out/ubench_$tag
mkdir -p "$out"
root=$PWD
cd /tmp && export TMPDIR=/tmp && cd "$root"
[ -x tools/bin/ubench_valu ] || { mkdir -p tools/bin; hipcc --offload-arch=gfx950 -O3 tools/ubench_valu.hip -o tools/bin/ubench_valu; }
tools/bin/ubench_valu > "$out/table.txt"
rm -rf "$out/pmc"
rocprofv3 --pmc SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc" -- \
    tools/bin/ubench_valu --waves 1,2,4,8 "add ind" "fma_f32 ind" "pk_min_u16 ind" "sweep opcode mix" "slow 1 in 32" "fast opcodes mixed ind" "lshl ind" > "$out/pmc_run.txt" 2>&1
python3 tools/ubench_pmc.py "$out/pmc" > "$out/pmc.txt"
rm -rf "$out/pmc"
echo done
