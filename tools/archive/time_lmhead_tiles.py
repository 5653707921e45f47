import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
M, N, Kd = 32768, 151936, 1024
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
a, b = r(M, Kd), r(N, Kd)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def bench(fn, n=6):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for rnd in range(2):
    for name, fn in (("tile 2", lambda: K.gemm(L.GEMM_NT, a, b, out=out, tile=2, allow_split_k=False)), ("tile 5", lambda: K.gemm(L.GEMM_NT, a, b, out=out, tile=5, allow_split_k=False)),
                     ("tile 1", lambda: K.gemm(L.GEMM_NT, a, b, out=out, tile=1, allow_split_k=False)), ("tile 3", lambda: K.gemm(L.GEMM_NT, a, b, out=out, tile=3, allow_split_k=False)), ("tile 4", lambda: K.gemm(L.GEMM_NT, a, b, out=out, tile=4, allow_split_k=False)),
                     ("library", lambda: torch.matmul(a, b.t(), out=out))):
        t = bench(fn)
        print(f"round {rnd}: {name:10s} {t:7.3f} ms  {2.0 * M * N * Kd / t / 1e9:7.0f} TFLOP/s", flush=True)
