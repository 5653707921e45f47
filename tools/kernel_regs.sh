#!/bin/bash
# Register / scratch use of one kernel of gemm.hip part N:  bash tools/kernel_regs.sh <part> <kernel name substring> [extra flags]  (device asm kept at /tmp/gemm_p<part>.s)
R=$(cd "$(dirname "$0")/.." && pwd); part=$1; pat=$2; shift 2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -DGEMM_PART=$part "$@" -I$R/include -S --cuda-device-only $R/llm_quest_amd/csrc/gemm.hip -o /tmp/gemm_p$part.s 2>&1 | grep -v "hip-link"
for n in $(grep -n "\.name:.*$pat" /tmp/gemm_p$part.s | cut -d: -f1); do sed -n "$((n-12)),$((n+22))p" /tmp/gemm_p$part.s | grep -E "\.name|vgpr_count|private_segment|vgpr_spill|group_segment_fixed"; done
