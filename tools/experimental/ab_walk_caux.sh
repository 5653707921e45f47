#!/bin/bash
# In-step: output-store cache policy of the weight-stationary-walk kernels (build_variants/libmi355vlm_wcaux{2,16,18}.so) with the walk by shape (2) and everywhere (1)
B="python bench.py --steps 6 --warmup 2 --cpu-baseline off --pipe-probe off --optimizer off --fp32-tower-leg off --other-configs off"
run() { MI355_GEMM_WALK=$2 MI355_LIB_PATH=$1 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3 walk=$2', d['ms_per_step'])"; }
for r in 1 2; do
run llm_quest_amd/libmi355vlm.so 2 base
run build_variants/libmi355vlm_wcaux2.so 2 nt
run build_variants/libmi355vlm_wcaux16.so 2 sc1
run build_variants/libmi355vlm_wcaux18.so 2 sc1nt
run llm_quest_amd/libmi355vlm.so 1 base
run build_variants/libmi355vlm_wcaux2.so 1 nt
run build_variants/libmi355vlm_wcaux16.so 1 sc1
run build_variants/libmi355vlm_wcaux18.so 1 sc1nt
done
