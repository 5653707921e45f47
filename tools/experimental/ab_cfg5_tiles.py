"""Config 5's N = 1024 projections (22 784 rows: 356 tiles of 256 x 256 on 256 CUs = 1.39 rounds): tile hints and split-K by HIP events.  python tools/experimental/ab_cfg5_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
M, N = 22784, 1024
for Kd in (1024, 2048, 3584, 6144):
    a, b = r(M, Kd), r(N, Kd)
    for tile, sk in ((0, False), (1, False), (2, False), (3, False), (2, True), (1, True)):
        try:
            out = K.gemm(L.GEMM_NT, a, b, tile=tile, allow_split_k=sk)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            for _ in range(3): K.gemm(L.GEMM_NT, a, b, out=out, tile=tile, allow_split_k=sk)
            s.record()
            for _ in range(30): K.gemm(L.GEMM_NT, a, b, out=out, tile=tile, allow_split_k=sk)
            e.record(); torch.cuda.synchronize()
            us = s.elapsed_time(e) / 30 * 1e3
            print(f"K {Kd:5d} tile {tile} split-k {int(sk)}: {us:8.1f} us  {2.0 * M * N * Kd / us / 1e6:7.1f} TFLOP/s", flush=True)
        except Exception as ex:
            print(f"K {Kd} tile {tile} split-k {sk}: {ex}")
