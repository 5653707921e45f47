# Tile-walk experiment (GPU box): memory-side traffic and duration of the gate-up forward GEMM under each super-group height.
#   bash tools/collect_walk.sh   -> gpurun_out/walk/{g3,g2,g4,g6,g8,g16}_{fetch,write}
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/walk; rm -rf $O; mkdir -p $O
M=45376
(cd $R && python3 -m llm_quest_amd.fingerprint ${GIT_SHA:-unknown} 64 > $O/stamp.json)
for spec in "g3 2" "g2 24578" "g4 8194" "g6 32770" "g8 40962" "g16 16386"; do
  set -- $spec
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${1}_fetch -- python3 $R/tools/gemm_one.py NT $2 $M 6144 1024 6 > $O/${1}_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/${1}_write -- python3 $R/tools/gemm_one.py NT $2 $M 6144 1024 6 > $O/${1}_write.log 2>&1
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/${1}_time -- python3 $R/tools/gemm_one.py NT $2 $M 6144 1024 40 > $O/${1}_time.log 2>&1
  echo $1 done
done
