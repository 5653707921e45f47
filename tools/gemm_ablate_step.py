"""Ablation switches of the per-tile NT kernel (tile 2: bit 0 = no DMA after the first K-tile, bit 1 = no fragment reads, bit 2 = no barrier) on the step's QKV shape:
what the main loop would run at without its memory side, its LDS side, its barriers.  GPU box only."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
M, N, Kd = 113440, 4096, int(sys.argv[1]) if len(sys.argv) > 1 else 1024
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
a, b = r(M, Kd), r(N, Kd)
out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
for ab in (0, 1, 2, 4, 3, 5, 7):
    tile = 2 | (ab << 8)
    for _ in range(3): K.gemm(L.GEMM_NT, a, b, out=out, tile=tile, allow_split_k=False)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): K.gemm(L.GEMM_NT, a, b, out=out, tile=tile, allow_split_k=False)
    e.record(); torch.cuda.synchronize()
    us = s.elapsed_time(e) / 10 * 1e3
    print(f"ablate {ab} ({'no DMA ' if ab & 1 else ''}{'no fragment reads ' if ab & 2 else ''}{'no barrier' if ab & 4 else ''}): {us:7.1f} us  {2.0 * M * N * Kd / us / 1e6:6.0f} TFLOP/s", flush=True)
