"""LM-head forward GEMM (NT, M = 32768 loss rows, N = 151936, K = 1024) under the tile walk's super-group heights.  usage: python tools/time_lmhead_walk.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
M, N, Kd = 32768, 151936, 1024
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
a, b = r(M, Kd), r(N, Kd)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
def bench(tile, n=6):
    for _ in range(2): K.gemm(L.GEMM_NT, a, b, out=out, tile=tile, allow_split_k=False)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): K.gemm(L.GEMM_NT, a, b, out=out, tile=tile, allow_split_k=False)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
for rnd in range(2):
    for name, tile in (("6 rows (default)", 2), ("3", 2 + (6 << 13)), ("4", 2 + (1 << 13)), ("8", 2 + (5 << 13)), ("16", 2 + (2 << 13))):
        t = bench(tile)
        print(f"round {rnd}: {name:18s} {t:7.3f} ms  {2.0 * M * N * Kd / t / 1e9:7.0f} TFLOP/s", flush=True)
