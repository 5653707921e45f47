# Evidence collection (GPU box, from the repo root):   ROUND=r05 GIT_SHA=<sha> BATCH=160 bash tools/collect_evidence.sh
# Writes raw rocprofv3 output under gpurun_out/${ROUND}ev/ and a stamp (kernel-source fingerprint, sha-256 of the loaded library, git sha, batch)
# taken HERE, at collection time; tools/aggregate_evidence.sh turns it into profiles/${ROUND}_* in the container and copies the stamp.
set -e
ROUND=${ROUND:-r06}
cd /tmp && export TMPDIR=/tmp
PART=${PART:-all}  # all | headline | configs  (a gpurun call is limited to 20 minutes: run the two halves as two calls; "configs" adds to the same directory)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${ROUND}ev; [ "$PART" = configs ] || rm -rf $O; mkdir -p $O  # (in the container: delete gpurun_out/${ROUND}ev before the call too -- gpurun MERGES what comes back into what is there)
B=${BATCH:-160}; M=$((B * 709))
(cd $R && python3 -m llm_quest_amd.fingerprint ${GIT_SHA:-unknown} $B > $O/stamp.json)
if [ "$PART" != configs ]; then
SQ1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -- python3 $R/bench.py --batch $B --steps 4 --warmup 2 --cpu-baseline off --pipe-probe off --fp32-tower-leg off --other-configs off > $O/bench_trace.log 2>&1
echo trace done
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/step_fetch -- python3 $R/bench.py --batch $B --steps 1 --warmup 1 --cpu-baseline off --pipe-probe off --fp32-tower-leg off --other-configs off > $O/step_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/step_write -- python3 $R/bench.py --batch $B --steps 1 --warmup 1 --cpu-baseline off --pipe-probe off --fp32-tower-leg off --other-configs off > $O/step_write.log 2>&1
echo step pmc done
for spec in "NT $M 6144 1024 nt_gateup" "NT $M 1024 6144 nt_dgrad"; do
  set -- $spec
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${5}_fetch -- python3 $R/tools/gemm_one.py $1 0 $2 $3 $4 3 > $O/${5}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${5}_write -- python3 $R/tools/gemm_one.py $1 0 $2 $3 $4 3 > $O/${5}_write.log 2>&1
  rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/${5}_sq1 -- python3 $R/tools/gemm_one.py $1 0 $2 $3 $4 6 > $O/${5}_sq1.log 2>&1
  rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/${5}_sq2 -- python3 $R/tools/gemm_one.py $1 0 $2 $3 $4 6 > $O/${5}_sq2.log 2>&1
  echo $5 done
done
# the block's four weight gradients as the STEP launches them: one grouped TN launch of 240 tiles (gemm_grouped_kernel)
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/tn_group_fetch -- python3 $R/tools/wgrad_group_one.py $B 3 > $O/tn_group_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/tn_group_write -- python3 $R/tools/wgrad_group_one.py $B 3 > $O/tn_group_write.log 2>&1
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/tn_group_sq1 -- python3 $R/tools/wgrad_group_one.py $B 4 > $O/tn_group_sq1.log 2>&1
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/tn_group_sq2 -- python3 $R/tools/wgrad_group_one.py $B 4 > $O/tn_group_sq2.log 2>&1
echo tn_group done
rocprofv3 --pmc $SQ1 --kernel-trace --output-format csv -d $O/attn_sq1 -- python3 $R/tools/attn_one.py $B 3 > $O/attn_sq1.log 2>&1
rocprofv3 --pmc $SQ2 --kernel-trace --output-format csv -d $O/attn_sq2 -- python3 $R/tools/attn_one.py $B 3 > $O/attn_sq2.log 2>&1
echo attn done
fi
if [ "$PART" != headline ]; then
# the other BASELINE configurations: whole-step memory-side traffic + kernel table of `bench.py --config N` (same passes as the headline's)
for cfg in 2 3 5; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/cfg${cfg}_trace -- python3 $R/bench.py --config $cfg --steps 3 --warmup 2 --cpu-baseline off > $O/cfg${cfg}_trace.log 2>&1
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/cfg${cfg}_fetch -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --cpu-baseline off > $O/cfg${cfg}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/cfg${cfg}_write -- python3 $R/bench.py --config $cfg --steps 1 --warmup 1 --cpu-baseline off > $O/cfg${cfg}_write.log 2>&1
  echo config $cfg done
done
fi
