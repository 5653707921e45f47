"""Summarise a rocprofv3 --kernel-trace --stats kernel_stats.csv: python tools/prof_summary.py <csv> <steps> [top]"""
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]); top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print(f"total kernel time per step: {tot/steps/1e6:.1f} ms")
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"TileCfg<256, 256, 2, 4, 32, 4>", "T256alt", n); n = re.sub(r"TileCfg<128, 128, 2, 2, 64, 2>", "T128", n); n = re.sub(r"TileCfg<256, 256, 2, 4, 64, 2>", "T256", n)
    n = n.replace("void ", "").replace("(GemmParams)", "").replace("(GroupTable)", "")
    return n[:70]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:top]:
    t = float(r["TotalDurationNs"]); c = int(r["Calls"])
    print(f"{short(r['Name']):72s} {c/steps:7.1f} {t/steps/1e6:8.2f} ms {t/c/1e3:9.1f} us {100*t/tot:5.1f}%")
