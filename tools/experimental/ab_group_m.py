"""Row-group height of the persistent NT walk on the LM head's shapes (forward [81 920, 151 936] = X W^T, K 1 024; dgrad [81 920, 1 024], K 151 936) and on the gate-up forward:
python tools/experimental/ab_group_m.py   (MI355_LIB_PATH selects the build; prints us per launch by HIP events)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
shapes = {"head fwd": (81920, 151936, 1024), "head dgrad": (81920, 1024, 151936), "gate-up fwd": (113440, 6144, 1024), "gate-up dgrad": (113440, 1024, 6144)}
for name, (M, N, Kd) in shapes.items():
    a, b = r(M, Kd), r(N, Kd)
    out = K.gemm(L.GEMM_NT, a, b)
    n = 3 if M * N * Kd > 5e12 else 20
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(2): K.gemm(L.GEMM_NT, a, b, out=out)
    s.record()
    for _ in range(n): K.gemm(L.GEMM_NT, a, b, out=out)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / n * 1e3
    print(f"{name:14s} {us:10.1f} us  {2.0 * M * N * Kd / us / 1e6:7.1f} TFLOP/s", flush=True)
    del a, b, out
