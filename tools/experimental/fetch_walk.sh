#!/bin/bash
# FETCH_SIZE of the persistent NT kernel with (hint 7 + 2048) and without (hint 7) the weight-stationary walk, per shape (GPU box, repo root).
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetch_walk; rm -rf $O; mkdir -p $O
B=${BATCH:-160}; M=$((B * 709)); MH=$((B * 512))
for spec in "qkv $M 4096 1024" "down $M 1024 3072" "dqkv $M 1024 4096" "head_fwd $MH 151936 1024" "head_dgrad $MH 1024 151936"; do
  set -- $spec
  for t in 7 2055; do
    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/${1}_$t -- python3 $R/tools/gemm_one.py NT $t $2 $3 $4 3 > $O/${1}_$t.log 2>&1
  done
  echo $1 done
done
python3 - <<PY
import csv, glob
for name in ("qkv", "down", "dqkv", "head_fwd", "head_dgrad"):
    row = []
    for t in (7, 2055):
        vals, durs = [], []
        for f in glob.glob("$O/%s_%d/**/*counter_collection.csv" % (name, t), recursive=True):
            for r in csv.DictReader(open(f)):
                if r["Counter_Name"] == "FETCH_SIZE" and "gemm_nt_persist" in r["Kernel_Name"]:
                    vals.append(float(r["Counter_Value"]))
        for f in glob.glob("$O/%s_%d/**/*kernel_trace.csv" % (name, t), recursive=True):
            for r in csv.DictReader(open(f)):
                if "gemm_nt_persist" in r["Kernel_Name"]:
                    durs.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        vals, durs = vals[1:] or vals, durs[1:] or durs
        row.append((2 * 1024 * sum(vals) / max(len(vals), 1) / 1e9, sum(durs) / max(len(durs), 1)))
    print("%-11s old walk %7.2f GB %9.1f us   weight-stationary %7.2f GB %9.1f us" % (name, row[0][0], row[0][1], row[1][0], row[1][1]))
PY
