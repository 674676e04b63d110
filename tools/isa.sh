#!/bin/bash
# tools/isa.sh <file.hip under csrc/> <out.s>: gfx950 ISA of one kernel file, then VGPR / scratch per kernel.
set -e
src=/root/repo/avisynth_sangnom2_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-gpu-flush-denormals-to-zero -fno-slp-vectorize \
    -I/root/repo/include -I$src -S --cuda-device-only $src/$1 -o $2 2>&1 | grep -v "argument unused" || true
grep -E "\.num_vgpr, |\.private_seg_size, " $2 | sed -E 's/.*\.set _ZN2sn[0-9a-z]*[0-9]+(k_[a-z0-9_]+)(ILi[0-9]+ELi[0-9]+E)?.*\.(num_vgpr|private_seg_size), ([0-9]+)/\1 \2 \3 \4/' | paste - - | awk '{print $1,$2,"vgpr",$4,"scratch",$8}'
