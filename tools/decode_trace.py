"""Per-kernel durations of the steady part of a rocprofv3 --kernel-trace of tools/bench_decode.py:  python tools/decode_trace.py <trace dir>"""
import collections
import csv
import glob
import sys

f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[int(len(rows) * 0.6):]  # the replayed steps
dur = collections.defaultdict(list)
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(rows, rows[1:])]
for r in rows:
    dur[r["Kernel_Name"].replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:44]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in dur.values())
wall = int(rows[-1]["End_Timestamp"]) - int(rows[0]["Start_Timestamp"])
print(f"dispatches {len(rows)}  kernel time {tot / 1e3:.0f} us  wall {wall / 1e3:.0f} us  gap median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us")
for k, v in sorted(dur.items(), key=lambda x: -sum(x[1]))[:12]:
    v.sort()
    print(f"{k:46s} n={len(v):5d}  min {v[0] / 1e3:6.2f}  p50 {v[len(v) // 2] / 1e3:6.2f}  p90 {v[int(len(v) * 0.9)] / 1e3:6.2f}  max {v[-1] / 1e3:7.2f} us  share {sum(v) / tot * 100:5.1f} %")
