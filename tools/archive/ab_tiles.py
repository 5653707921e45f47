"""Same-process A/B of GEMM tile configurations on the step's NT shapes (sustained launches, HIP events on the launch stream).
usage: python tools/ab_tiles.py [tile hints ...]   (0 = the step's default for the form)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
tiles = [int(a) for a in sys.argv[1:]] or [0, 1, 2, 3, 5]
M = 64 * 709
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
shapes = {"gate-up (N 6144, K 1024)": (6144, 1024), "QKV (N 4096, K 1024)": (4096, 1024), "dctx dgrad (N 2048, K 1024)": (2048, 1024),
          "down (N 1024, K 3072)": (1024, 3072), "out-proj (N 1024, K 2048)": (1024, 2048), "gate-up dgrad (N 1024, K 6144)": (1024, 6144)}
def bench(fn, n=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, (N, Kd) in shapes.items():
    x, w = r(M, Kd), r(N, Kd)
    out = []
    for t in tiles:
        best = min(bench(lambda: K.gemm(L.GEMM_NT, x, w, tile=t)) for _ in range(2))
        out.append(f"tile {t}: {best:7.1f} us {2.0 * M * N * Kd / best / 1e6:6.0f} TF")
    print(f"{name:32s} " + "   ".join(out))
