"""Per-kernel SQ / GRBM counters from rocprofv3 --pmc passes:  python tools/pmc_sq.py <out.json> <label> <collection_dir> <dir> [<dir> ...]
Each <dir> is the output of one `rocprofv3 --pmc ... --kernel-trace` pass over the same program (slots: 8 SQ + 2 GRBM per pass).
Counters are averaged per dispatch and merged per kernel; derived columns follow MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES cycles (32 per v_mfma_f32_32x32x16_bf16, 16 per 16x16x32);
clock = GRBM_GUI_ACTIVE / 8 XCDs / duration."""
import csv, glob, json, re, subprocess, sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    return re.sub(r"\(.*", "", n)[:100]


out_path, label, coll, dirs = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
agg = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
dur = defaultdict(lambda: [0, 0.0])
for d in dirs:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            a = agg[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            t = dur[short(r["Kernel_Name"])]
            t[0] += 1
            t[1] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from llm_quest_amd.fingerprint import collection_stamp
stamp = collection_stamp(coll)  # written on the GPU box when the counters were collected
sha = stamp["git_sha"]
res = {"label": label, "git_sha": sha, "note": __doc__.split("\n\n")[0].split("\n", 1)[1] if False else "per-dispatch averages; see tools/pmc_sq.py for units", "kernels": {}}
for k, cs in agg.items():
    row = {c: v[1] / v[0] for c, v in cs.items()}
    if k in dur and dur[k][0]:
        row["avg_duration_us_under_pmc"] = dur[k][1] / dur[k][0] / 1e3
    wc, mf = row.get("SQ_WAVE_CYCLES"), row.get("SQ_VALU_MFMA_BUSY_CYCLES")
    if row.get("GRBM_GUI_ACTIVE") and "avg_duration_us_under_pmc" in row:
        row["clock_GHz"] = row["GRBM_GUI_ACTIVE"] / 8 / (row["avg_duration_us_under_pmc"] * 1e3)
    if mf and row.get("SQ_BUSY_CYCLES"):
        pass
    if mf and row.get("GRBM_GUI_ACTIVE"):
        # matrix-pipe busy share of the chip: MFMA-busy cycles summed over SIMDs / (1024 SIMDs x active cycles per XCD)
        row["mfma_busy_pct"] = 100.0 * mf / (1024 * row["GRBM_GUI_ACTIVE"] / 8)
    if wc:
        for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS"):
            if c in row:
                row[c + "_share_of_wave_cycles"] = row[c] / wc
    if row.get("SQ_INSTS_MFMA") and row.get("SQ_INSTS_VALU"):
        row["valu_per_mfma"] = (row["SQ_INSTS_VALU"] - row["SQ_INSTS_MFMA"]) / row["SQ_INSTS_MFMA"] if row["SQ_INSTS_VALU"] > row["SQ_INSTS_MFMA"] else row["SQ_INSTS_VALU"] / row["SQ_INSTS_MFMA"]
    res["kernels"][k] = row
try:
    allres = json.load(open(out_path))
except (OSError, ValueError):
    allres = {"runs": []}
allres["runs"] = [r for r in allres["runs"] if r["label"] != label] + [res]
allres["kernel_sources_sha"], allres["library_sha"] = stamp["kernel_sources_sha"], stamp["library_sha"]
json.dump(allres, open(out_path, "w"), indent=1)
for k, row in sorted(res["kernels"].items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:12]:
    print(k[:70], {c: (round(v, 3) if v < 100 else int(v)) for c, v in row.items() if c.endswith(("pct", "GHz", "cycles", "mfma", "pmc"))})
