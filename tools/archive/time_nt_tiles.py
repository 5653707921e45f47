import sys, os, torch
sys.path.insert(0, "/root/repo")
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
M = 113440
def timed(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for name, (N, Kd) in (("QKV", (4096, 1024)), ("N1024 K4096", (1024, 4096)), ("N3072 K1024", (3072, 1024)), ("N1024 K1024", (1024, 1024))):
    x, w = r(M, Kd), r(N, Kd); o = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    line = []
    for rep in range(2):
        for tile in (2, 3, 4, 5, 7):
            t = timed(lambda: K.gemm(L.GEMM_NT, x, w, out=o, tile=tile, allow_split_k=False))
            line.append(f"t{tile} {t:6.1f}")
    print(name, " | ".join(line), flush=True)
