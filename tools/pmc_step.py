"""Whole-step memory-side traffic from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, each with --kernel-trace) over
`python3 bench.py --steps 1 --warmup 1 --cpu-baseline off`:  python tools/pmc_step.py <fetch_dir> <write_dir> <steps_in_trace> <out.json> <collection_dir>
Units and corrections as MI355X_MICROARCH.md prescribes: rocprofv3 reports KiB; on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes, so it is doubled."""
import csv, glob, json, os, re, subprocess, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd.fingerprint import collection_stamp


def load(d, counter):
    per = defaultdict(lambda: [0, 0.0])
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = per[r["Kernel_Name"]]
            k[0] += 1
            k[1] += float(r["Counter_Value"])
    return per


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    return re.sub(r"\(.*", "", n)[:110]


fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
steps = float(sys.argv[3])
rows = []
for name in set(fetch) | set(write):
    n = max(fetch.get(name, [0])[0], write.get(name, [0])[0])
    per_step = int(n // steps)  # bench.py also times the dominant GEMM alone (13 launches per form): those fall out with the remainder
    if per_step == 0 or "adamw" in name:  # the headline step has no optimizer in it
        continue
    fb = 2.0 * 1024.0 * fetch.get(name, [0, 0.0])[1] / n * per_step
    wb = 1024.0 * write.get(name, [0, 0.0])[1] / n * per_step
    rows.append({"kernel": short(name), "launches_per_step": per_step, "fetch_bytes": int(fb), "write_bytes": int(wb)})
rows.sort(key=lambda r: -(r["fetch_bytes"] + r["write_bytes"]))
tot_f, tot_w = sum(r["fetch_bytes"] for r in rows), sum(r["write_bytes"] for r in rows)
stamp = collection_stamp(sys.argv[5])  # written on the GPU box when the counters were collected
config = sys.argv[6] if len(sys.argv) > 6 else None  # "config2" / "config3" / "config5": `bench.py --config N` at its default batch (no optimizer re-run: warm-up + timed steps only)
out = {
    "kernel_sources_sha": stamp["kernel_sources_sha"], "all_sources_sha": stamp.get("all_sources_sha"), "library_sha": stamp["library_sha"], "git_sha": stamp["git_sha"],
    "per_gpu_batch": {"config2": 332, "config3": 64, "config5": 32}[config] if config else stamp["per_gpu_batch"], **({"config": config} if config else {}),
    "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes, TCC slots) with --kernel-trace over `python3 bench.py --steps 1 --warmup 1 "
            "--cpu-baseline off` (per-GPU batch as stamped: 4 steps in each trace -- warm-up + timed, without and with the optimizer); figures are PER STEP = per-launch "
            "average x launches per step, so the bench's 13 stand-alone launches of each gate-up GEMM form drop out, and the optimizer's own kernels are left out (the headline step has none).  KiB units from rocprofv3; gfx950 correction: FETCH_SIZE doubled (128-byte requests "
            "tallied at 64), WRITE_SIZE exact.  Infinity-Cache hits are included in FETCH_SIZE (memory-side L2 requests).",
    "per_step": {"fetch_bytes": tot_f, "write_bytes": tot_w, "total_bytes": tot_f + tot_w},
    "kernels": rows[:40],
}
json.dump(out, open(sys.argv[4], "w"), indent=1)
print(json.dumps(out["per_step"]))
