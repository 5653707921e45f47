"""Aggregate a rocprofv3 --kernel-trace CSV by (kernel, grid size): calls, average / total time per step.
    python tools/trace_by_shape.py <kernel_trace.csv> <steps> [top]
GEMM launches of one tile configuration differ only in their grid (= tile count), so this separates the shapes of a step."""
import csv, re, sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 60
agg = defaultdict(lambda: [0, 0.0])
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n).replace("void ", "")
    n = re.sub(r"TileCfg<256, 256, 2, 4, 32, 4>", "T3", n); n = re.sub(r"TileCfg<128, 128, 2, 2, 64, 2>", "T1", n); n = re.sub(r"TileCfg<256, 256, 2, 4, 64, 2>", "T2", n)
    return re.sub(r"\(.*", "", n)[:60]
for r in rows:
    name = r.get("Kernel_Name") or r.get("Name")
    gx = int(r.get("Grid_Size_X", r.get("Grid_Size", 0))); wg = int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 1)) or 1)
    dur = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    k = (short(name), gx // max(wg, 1))
    agg[k][0] += 1; agg[k][1] += dur
tot = sum(v[1] for v in agg.values())
print(f"total kernel time per step {tot/steps/1e6:.2f} ms")
for (name, blocks), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{name:62s} blocks {blocks:7d} calls/step {c/steps:7.1f} avg {t/c/1e3:9.1f} us  {t/steps/1e6:8.2f} ms/step {100*t/tot:5.1f}%")
