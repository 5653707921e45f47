"""Mean duration of the attention backward kernels in a rocprofv3 kernel trace of tools/ab_attn_bwd.py, split by which dQ form followed the dK/dV pass."""
import collections, csv, glob, sys
f = glob.glob((sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/ab") + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
seq = [(r["Kernel_Name"], (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3) for r in rows]
short = lambda n: n.split("::")[-1].split("(")[0] if "attn" in n else None
d = collections.defaultdict(list)
for i, (n, t) in enumerate(seq):
    s = short(n)
    if s is None: continue
    if "dkv" in s: s += " -> " + (short(seq[i + 1][0]) or "?")
    d[s].append(t)
for k, v in d.items(): print(f"{k:60s} n={len(v):3d} mean={sum(v)/len(v):8.1f} min={min(v):8.1f}")
