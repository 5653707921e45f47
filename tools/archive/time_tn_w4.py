"""Weight-gradient (TN) GEMMs of the step at batch 160 (K = 113 440 tokens): the 8-wave alternating loop (tile 3, the default) against the 4-wave 128x128-per-wave
loop (tile 5) -- single launches, the grouped launch of a block's four weight gradients, and the LM head's.  Results must be bit-identical.  GPU box only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
Kd = int(sys.argv[1]) if len(sys.argv) > 1 else 113440


def timed(fn, n):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


shapes = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]
ops = [(r(Kd, M), r(Kd, N), torch.empty(M, N, dtype=torch.bfloat16, device="cuda")) for M, N in shapes]
flop = sum(2.0 * M * N * Kd for M, N in shapes)
outs = {}
for tile in (3, 5, 3, 5):
    us = timed(lambda: K.gemm_grouped(L.GEMM_TN, [(a, b, o, None) for a, b, o in ops], tile=tile), 10)
    outs[tile] = [o.clone() for _, _, o in ops]
    print(f"grouped block wgrads K={Kd} tile {tile}: {us:8.1f} us  {flop / us / 1e6:7.1f} TFLOP/s", flush=True)
print("bit-identical tile 3 vs 5:", all(torch.equal(x, y) for x, y in zip(outs[3], outs[5])))
for (M, N), (a, b, o) in zip(shapes, ops):
    for tile in (3, 5):
        us = timed(lambda: K.gemm(L.GEMM_TN, a, b, out=o, tile=tile, allow_split_k=False), 10)
        print(f"single dW[{M},{N}] tile {tile}: {us:8.1f} us  {2.0 * M * N * Kd / us / 1e6:7.1f} TFLOP/s", flush=True)
del ops
Kh = 81920
a, b = r(Kh, 151936), r(Kh, 1024)
o = torch.empty(151936, 1024, dtype=torch.bfloat16, device="cuda")
for tile in (3, 5, 3, 5):
    us = timed(lambda: K.gemm(L.GEMM_TN, a, b, out=o, tile=tile, allow_split_k=False), 4)
    print(f"LM-head dW[151936,1024] K={Kh} tile {tile}: {us:8.1f} us  {2.0 * 151936 * 1024 * Kh / us / 1e6:7.1f} TFLOP/s", flush=True)
