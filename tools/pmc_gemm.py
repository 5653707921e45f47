"""Memory-side traffic of the step's dominant GEMM, per form, from rocprofv3 --pmc passes over tools/gemm_one.py (one GEMM per process):
    python tools/pmc_gemm.py <out.json> <collection_dir> FORM:<fetch_dir>:<write_dir>:M:N:K:what ...
FETCH_SIZE / WRITE_SIZE are collected in SEPARATE passes (TCC slots); rocprofv3 reports KiB; gfx950 correction: FETCH_SIZE doubled
(128-byte requests tallied at 64), WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section).  The first launch of a process is cold and is dropped."""
import csv, glob, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd.fingerprint import collection_stamp


def avg(d, counter):
    vals, durs = [], []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and ("gemm_bf16_kernel" in r["Kernel_Name"] or "gemm_nt_persist_kernel" in r["Kernel_Name"]):
                vals.append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if ("gemm_bf16_kernel" in r["Kernel_Name"] or "gemm_nt_persist_kernel" in r["Kernel_Name"]):
                durs.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    vals, durs = vals[1:] or vals, durs[1:] or durs
    return sum(vals) / len(vals), sum(durs) / len(durs)


stamp = collection_stamp(sys.argv[2])  # written on the GPU box when the counters were collected
out = {"kernel_sources_sha": stamp["kernel_sources_sha"], "library_sha": stamp["library_sha"], "git_sha": stamp["git_sha"],
       "note": __doc__.strip().split("\n", 2)[2], "kernels": {}}
for spec in sys.argv[3:]:
    form, fd, wd, M, N, K, what = spec.split(":", 6)
    M, N, K = int(M), int(N), int(K)
    f_kib, f_us = avg(fd, "FETCH_SIZE")
    w_kib, w_us = avg(wd, "WRITE_SIZE")
    rd, wr = 2 * (M * K + N * K), 2 * M * N
    if form == "TN":
        rd, wr = 2 * (K * M + K * N), 2 * M * N
    fb, wb = int(2 * 1024 * f_kib), int(1024 * w_kib)
    out["kernels"][form] = {"what": what, "M": M, "N": N, "K": K, "algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr,
                            "FETCH_SIZE_KiB_per_launch": round(f_kib, 1), "avg_us_under_FETCH_SIZE": round(f_us, 1),
                            "WRITE_SIZE_KiB_per_launch": round(w_kib, 1), "avg_us_under_WRITE_SIZE": round(w_us, 1),
                            "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "over_algorithmic": round((fb + wb) / (rd + wr), 2)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes"], v["over_algorithmic"]) for k, v in out["kernels"].items()}))
