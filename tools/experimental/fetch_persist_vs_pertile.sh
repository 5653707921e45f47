cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fetchcmp; rm -rf $O; mkdir -p $O
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/persist -- python3 $R/tools/gemm_one.py NT 0 113440 6144 1024 3 > $O/p.log 2>&1
MI355_GEMM_PERSIST_MIN_TILES=100000000 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pertile -- python3 $R/tools/gemm_one.py NT 0 113440 6144 1024 3 > $O/t.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/persist_head -- python3 $R/tools/gemm_one.py NT 0 81920 151936 1024 2 > $O/ph.log 2>&1
MI355_GEMM_PERSIST_MIN_TILES=100000000 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pertile_head -- python3 $R/tools/gemm_one.py NT 0 81920 151936 1024 2 > $O/th.log 2>&1
cd $R && python3 - <<'PY'
import csv,glob
for d in ("persist","pertile","persist_head","pertile_head"):
    f=glob.glob(f"gpurun_out/fetchcmp/{d}/*/*counter_collection.csv")[0]
    rows=[r for r in csv.DictReader(open(f)) if r["Counter_Name"]=="FETCH_SIZE" and "gemm" in r["Kernel_Name"]]
    for r in rows[-2:]:
        print(d, r["Kernel_Name"][28:70], "fetch GB", round(float(r["Counter_Value"])*2048/1e9,2), "us", (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
PY
