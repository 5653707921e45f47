"""This package's GEMM kernel vs the vendor library (torch.matmul -> hipBLASLt / rocBLAS) on the training step's own shapes, all three forms,
sustained (200 back-to-back launches on random data, after a warm-up burst).  A yardstick for the roofline fraction, not a dependency: the
product path never calls the library.  GPU box only: python tools/gemm_vs_lib.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from llm_quest_amd import _lib as L, kernels as K  # noqa: E402

T = 45376  # tokens per step at the bench's batch (64 x 709)
SHAPES = [("qkv", 4096, 1024), ("out_proj", 1024, 2048), ("gate_up", 6144, 1024), ("down", 1024, 3072), ("lm_head", 151936, 1024)]


def rate(fn, flops, n=200):
    for _ in range(20):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return flops * n / s.elapsed_time(e) / 1e9


print(f"{'shape':10s} {'form':5s} {'ours TF/s':>10s} {'lib TF/s':>10s} {'ours/lib':>8s}")
for name, N, Kd in SHAPES:
    M = 32768 if name == "lm_head" else T  # the head sees the 512 loss rows of each sample
    n = 40 if name == "lm_head" else 200
    x = torch.randn(M, Kd, device="cuda").bfloat16()
    w = (torch.randn(N, Kd, device="cuda") * 0.02).bfloat16()
    dy = torch.randn(M, N, device="cuda").bfloat16()
    fl = 2.0 * M * N * Kd
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    dx = torch.empty(M, Kd, device="cuda", dtype=torch.bfloat16)
    dw = torch.empty(N, Kd, device="cuda", dtype=torch.bfloat16)
    cases = [
        ("NT", lambda: K.gemm(L.GEMM_NT, x, w, out=y), lambda: torch.matmul(x, w.t(), out=y)),
        ("NN", lambda: K.gemm(L.GEMM_NN, dy, w, out=dx), lambda: torch.matmul(dy, w, out=dx)),
        ("TN", lambda: K.gemm(L.GEMM_TN, dy, x, out=dw), lambda: torch.matmul(dy.t(), x, out=dw)),
    ]
    for form, ours, lib in cases:
        time.sleep(1.0)
        a = rate(ours, fl, n)
        time.sleep(1.0)
        b = rate(lib, fl, n)
        print(f"{name:10s} {form:5s} {a:10.1f} {b:10.1f} {a / b:8.2f}", flush=True)
    del x, w, dy, y, dx, dw
    torch.cuda.empty_cache()
