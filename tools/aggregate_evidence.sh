#!/bin/bash
# Turn the raw output of tools/collect_evidence.sh (merged back under gpurun_out/r02ev/) into the tracked files under profiles/.
# Run from the repo root, in the container (no GPU needed).  Remove gpurun_out/r02ev before collecting: files of an older collection
# that are still there (other process ids) would be averaged in.
set -e
E=${1:-gpurun_out/r02ev}
for d in $E/*/; do n=$(ls $d*/ 2>/dev/null | sed 's/_.*//' | sort -u | wc -l); [ "$n" -le 1 ] || { echo "$d holds files of $n processes: stale collection mixed in"; exit 1; }; done
python tools/pmc_step.py $E/step_fetch $E/step_write 4 profiles/r02_pmc_tcc_step.json | tail -1
python tools/pmc_gemm.py profiles/r02_pmc_tcc_gemm.json \
  "NT:$E/nt_gateup_fetch:$E/nt_gateup_write:45376:6144:1024:gemm_bf16_kernel NT tile 2, gate-up forward C[45376,6144] = X[45376,1024] W[6144,1024]^T (plain epilogue)" \
  "NT_dgrad:$E/nt_dgrad_fetch:$E/nt_dgrad_write:45376:1024:6144:gemm_bf16_kernel NT tile 2, gate-up dgrad dX[45376,1024] = dY[45376,6144] (W^T)[1024,6144]^T (the step's form)" \
  "TN:$E/tn_wgrad_fetch:$E/tn_wgrad_write:6144:1024:45376:gemm_bf16_kernel TN tile 3, gate-up wgrad dW[6144,1024] = dY[45376,6144]^T X[45376,1024]" | tail -1
for l in nt_gateup nt_dgrad tn_wgrad attn; do python tools/pmc_sq.py profiles/r02_pmc_sq_counters.json ${l}_b64 $E/${l}_sq1 $E/${l}_sq2 > /dev/null; done
cp $(ls $E/trace/*/*kernel_stats.csv) profiles/r02_b_kernel_stats_b64.csv
python tools/trace_by_shape.py $(ls $E/trace/*/*kernel_trace.csv) 12 60 > profiles/r02_b_by_shape.txt
cp $E/bench_final.json profiles/r02_bench_b64.json
python - <<'PY'
import json
l = [x for x in open('profiles/r02_bench_b64.json') if x.startswith('{')][-1]
d = json.loads(l)
print("bench:", d['ms_per_step'], "ms/step", d['value'], d['unit'], "frac", d['roofline']['frac'], "with optimizer", d['with_optimizer_step']['ms_per_step'], "cpu", d['cpu_baseline']['value'])
PY
