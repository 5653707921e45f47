"""Sustained timing of the two GEMM launches that carry a fused SwiGLU (the step's shapes at per-GPU batch 64), HIP events on the launch stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
M = 64 * 709
r = lambda *s: (torch.randn(*s, device="cuda") * 0.5).to(torch.bfloat16)
x, w_gu, w_down, dy, gu = r(M, 1024), r(6144, 1024), r(1024, 3072), r(M, 1024), r(M, 6144)
def bench(fn, n=40):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
flop = 2.0 * M * 6144 * 1024
for name, fn, fl in (("gate-up fwd + SwiGLU", lambda: K.gemm_gateup_swiglu(x, w_gu), flop), ("gate-up fwd plain", lambda: K.gemm(L.GEMM_NT, x, w_gu), flop),
                     ("down dgrad + SwiGLU bwd", lambda: K.gemm_dgrad_swiglu_bwd(dy, w_down, gu), flop / 2), ("down dgrad plain", lambda: K.dgrad(dy, w_down), flop / 2)):
    t = min(bench(fn) for _ in range(3))
    print(f"{name:28s} {t:8.1f} us  {fl / t / 1e6:7.0f} TFLOP/s")
