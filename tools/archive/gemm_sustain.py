"""Burst vs sustained throughput of the gate-up forward GEMM (45376 x 6144 x 1024, NT), random vs zero operands: is the step limited by clocks
(ramp-up from idle, power give-back on real data)?  Measured: 5 launches 794, 50: 862, 500: 932, 3000: 941 TFLOP/s (tile 2, random data), 973 on zeros --
short isolated sweeps UNDER-state the kernels (clock ramp), sustained random data costs 3.4 % against zeros.  GPU box only: python tools/gemm_sustain.py"""
import sys, os, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from llm_quest_amd import _lib as L, kernels as K
M, N, Kd = 45376, 6144, 1024
a = torch.randn(M, Kd, device="cuda").bfloat16(); b = torch.randn(N, Kd, device="cuda").bfloat16()
z = torch.zeros_like(a); zb = torch.zeros_like(b)
out = K.gemm(L.GEMM_NT, a, b, tile=2)
fl = 2.0 * M * N * Kd
def run(x, y, n, tile):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): K.gemm(L.GEMM_NT, x, y, out=out, tile=tile)
    e.record(); torch.cuda.synchronize()
    return fl * n / s.elapsed_time(e) / 1e9
for tile in (2, 3):
    for n in (5, 50, 500, 3000):
        time.sleep(2.0)  # let the chip idle
        print(f"tile {tile} random data, {n:5d} back-to-back launches: {run(a, b, n, tile):7.1f} TFLOP/s")
    time.sleep(2.0)
    print(f"tile {tile} ZERO operands, 3000 launches: {run(z, zb, 3000, tile):7.1f} TFLOP/s")
