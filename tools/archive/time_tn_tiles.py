"""Weight-gradient (TN) GEMM shapes of the step per tile configuration, sustained.  GPU box only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
def run(M, N, Kd, tile, n=40):
    a, b = r(Kd, M), r(Kd, N)
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    for _ in range(5): K.gemm(L.GEMM_TN, a, b, out=out, tile=tile, allow_split_k=False)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): K.gemm(L.GEMM_TN, a, b, out=out, tile=tile, allow_split_k=False)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    return us, 2.0 * M * N * Kd / us / 1e6
for (M, N, Kd, n) in ((6144, 1024, 45376, 40), (4096, 1024, 45376, 40), (1024, 3072, 45376, 40), (1024, 2048, 45376, 40), (151936, 1024, 32832, 6)):
    for tile in (3, 2, 4, 3):
        try:
            us, tf = run(M, N, Kd, tile, n)
            print(f"TN dW[{M},{N}] K={Kd} tile {tile}: {us:8.1f} us  {tf:7.1f} TFLOP/s", flush=True)
        except Exception as ex:
            print(f"TN dW[{M},{N}] K={Kd} tile {tile}: {type(ex).__name__} {str(ex)[:100]}", flush=True)
