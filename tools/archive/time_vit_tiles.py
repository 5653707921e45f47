import sys, os, torch
sys.path.insert(0, "/root/repo")
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
def timed(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
M = 160 * 197
for (N, Kd, odt, res) in ((768, 768, torch.float32, True), (768, 3072, torch.float32, True), (2304, 768, torch.bfloat16, False), (3072, 768, torch.bfloat16, False)):
    x, w = r(M, Kd), r(N, Kd)
    bias = torch.randn(N, device="cuda")
    rs = torch.randn(M, N, device="cuda", dtype=odt) if res else None
    o = torch.empty(M, N, device="cuda", dtype=odt)
    line = []
    for tile in (1, 2, 3, 5):
        t = timed(lambda: K.gemm(L.GEMM_NT, x, w, out=o, out_dtype=odt, bias=bias, residual=rs, tile=tile, allow_split_k=False))
        line.append(f"tile {tile}: {t:6.1f} us {2.0*M*N*Kd/t/1e6:5.0f} TF")
    print(f"N {N} K {Kd} {str(odt)[6:]} res {res}: " + " | ".join(line), flush=True)
