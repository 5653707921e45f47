"""Memory-side traffic of the step's dominant GEMM, per form, from rocprofv3 --pmc passes over tools/gemm_one.py (one GEMM per process):
    python tools/pmc_gemm.py <out.json> <collection_dir> FORM:<fetch_dir>:<write_dir>:M:N:K:what ...
FETCH_SIZE / WRITE_SIZE are collected in SEPARATE passes (TCC slots); rocprofv3 reports KiB; gfx950 correction: FETCH_SIZE doubled
(128-byte requests tallied at 64), WRITE_SIZE exact (MI355X_MICROARCH.md, HBM section).  The first launch of a process is cold and is dropped."""
import csv, glob, json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd.fingerprint import collection_stamp


GEMM_KERNELS = ("gemm_bf16_kernel", "gemm_nt_persist_kernel", "gemm_grouped_kernel")


def avg(d, counter):
    vals, durs = [], []
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and any(k in r["Kernel_Name"] for k in GEMM_KERNELS):
                vals.append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if any(k in r["Kernel_Name"] for k in GEMM_KERNELS):
                durs.append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
    vals, durs = vals[1:] or vals, durs[1:] or durs
    return sum(vals) / len(vals), sum(durs) / len(durs)


stamp = collection_stamp(sys.argv[2])  # written on the GPU box when the counters were collected
out = {"kernel_sources_sha": stamp["kernel_sources_sha"], "library_sha": stamp["library_sha"], "git_sha": stamp["git_sha"],
       "note": __doc__.strip().split("\n", 2)[2], "kernels": {}}
for spec in sys.argv[3:]:
    form, fd, wd, M, N, K, what = spec.split(":", 6)
    Ms, Ns, K = [int(x) for x in M.split("+")], [int(x) for x in N.split("+")], int(K)  # a grouped launch lists its problems' M and N joined by "+"
    f_kib, f_us = avg(fd, "FETCH_SIZE")
    w_kib, w_us = avg(wd, "WRITE_SIZE")
    rd = sum(2 * (m * K + n * K) for m, n in zip(Ms, Ns))  # both operands once (NT: [M,K] and [N,K]; TN: [K,M] and [K,N])
    wr = sum(2 * m * n for m, n in zip(Ms, Ns))
    M, N = (Ms[0], Ns[0]) if len(Ms) == 1 else (Ms, Ns)
    fb, wb = int(2 * 1024 * f_kib), int(1024 * w_kib)
    out["kernels"][form] = {"what": what, "M": M, "N": N, "K": K, "algorithmic_read_bytes": rd, "algorithmic_write_bytes": wr,
                            "FETCH_SIZE_KiB_per_launch": round(f_kib, 1), "avg_us_under_FETCH_SIZE": round(f_us, 1),
                            "WRITE_SIZE_KiB_per_launch": round(w_kib, 1), "avg_us_under_WRITE_SIZE": round(w_us, 1),
                            "fetch_bytes_corrected": fb, "write_bytes": wb, "hbm_bytes": fb + wb, "over_algorithmic": round((fb + wb) / (rd + wr), 2)}
json.dump(out, open(sys.argv[1], "w"), indent=1)
print(json.dumps({k: (v["hbm_bytes"], v["over_algorithmic"]) for k, v in out["kernels"].items()}))
