"""The down-projection's dgrad with the SwiGLU backward in its epilogue (step shape: dY [113 440, 1 024] x W2^T -> d(gate-up) [113 440, 6 144]) by tile: does a 2-workgroups-per-CU
tile (hint 1: 128 x 128) hide the vector-heavy epilogue under the other workgroup's MFMAs?  python tools/experimental/ab_swiglu_bwd_tiles.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llm_quest_amd import kernels as K
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
M, F, N = 113440, 3072, 1024
dy, w, gu = r(M, N), r(N, F) * 0.03, r(M, 2 * F)
for tile in (0, 1, 2, 3, 7):
    try:
        K.gemm_dgrad_swiglu_bwd(dy, w, gu, tile=tile)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(20): K.gemm_dgrad_swiglu_bwd(dy, w, gu, tile=tile)
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) / 20 * 1e3
        print(f"tile {tile}: {us:8.1f} us  {2.0 * M * F * N / us / 1e6:7.1f} TFLOP/s", flush=True)
    except Exception as ex:
        print(f"tile {tile}: {str(ex)[:120]}")
