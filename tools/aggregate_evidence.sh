#!/bin/bash
# Turn the raw output of tools/collect_evidence.sh (merged back under gpurun_out/${ROUND}ev/) into the tracked files under profiles/.
# Run from the repo root, in the container (no GPU needed).  Every output carries the collection's own stamp (gpurun_out/${ROUND}ev/stamp.json).
set -e
ROUND=${ROUND:-r06}
E=${1:-gpurun_out/${ROUND}ev}
for d in $E/*/; do n=$(ls $d*/ 2>/dev/null | sed 's/_.*//' | sort -u | wc -l); [ "$n" -le 1 ] || { echo "$d holds files of $n processes: stale collection mixed in"; exit 1; }; done
B=$(python -c "import json;print(json.load(open('$E/stamp.json'))['per_gpu_batch'])"); M=$((B * 709))
python tools/pmc_step.py $E/step_fetch $E/step_write 4 profiles/${ROUND}_pmc_tcc_step.json $E | tail -1
python tools/pmc_gemm.py profiles/${ROUND}_pmc_tcc_gemm.json $E \
  "NT:$E/nt_gateup_fetch:$E/nt_gateup_write:$M:6144:1024:gemm_nt_persist_kernel (tile 2 as a persistent workgroup per CU), gate-up forward C[$M,6144] = X[$M,1024] W[6144,1024]^T (plain epilogue)" \
  "NT_dgrad:$E/nt_dgrad_fetch:$E/nt_dgrad_write:$M:1024:6144:gemm_nt_persist_kernel (tile 2 as a persistent workgroup per CU), gate-up dgrad dX[$M,1024] = dY[$M,6144] (W^T)[1024,6144]^T (the step's form)" \
  "TN:$E/tn_group_fetch:$E/tn_group_write:4096+1024+6144+1024:1024+2048+1024+3072:$M:gemm_grouped_kernel TN tile 5 (four waves of 128x128): the block's four weight gradients as the step launches them, ONE grouped launch of 240 tiles (QKV 4096x1024, out_proj 1024x2048, gate-up 6144x1024, down 1024x3072; K = $M tokens)" | tail -1
rm -f profiles/${ROUND}_pmc_sq_counters.json
for l in nt_gateup nt_dgrad tn_group attn; do python tools/pmc_sq.py profiles/${ROUND}_pmc_sq_counters.json ${l}_b$B $E $E/${l}_sq1 $E/${l}_sq2 > /dev/null; done
cp $(ls $E/trace/*/*kernel_stats.csv) profiles/${ROUND}_kernel_stats_b$B.csv
python tools/trace_by_shape.py $(ls $E/trace/*/*kernel_trace.csv) 12 60 > profiles/${ROUND}_by_shape_b$B.txt
cp $E/stamp.json profiles/${ROUND}_stamp.json
head -3 profiles/${ROUND}_by_shape_b$B.txt
for cfg in 2 3 5; do
  [ -d $E/cfg${cfg}_fetch ] || continue
  python tools/pmc_step.py $E/cfg${cfg}_fetch $E/cfg${cfg}_write 2 profiles/${ROUND}_pmc_tcc_step_config${cfg}.json $E config$cfg | tail -1
  python tools/trace_by_shape.py $(ls $E/cfg${cfg}_trace/*/*kernel_trace.csv) 5 30 > profiles/${ROUND}_by_shape_config${cfg}.txt
done
