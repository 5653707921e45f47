"""What a few busy CUs cost a GEMM launch: the per-tile kernel (hint 2) and the persistent kernel (hint 7, fixed share per workgroup) timed alone and beside an "occupier" on a second
stream that holds ~N CUs for the whole measurement (a 128x128-tile GEMM with few tiles and a very long K: two workgroups per CU, nothing else fits beside them).  This is the
situation of a backward whose bucket all-reduces run beside its GEMMs (RCCL: one workgroup = one CU per channel).  GPU box only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
M = 113440
x, w = r(M, 1024), r(4096, 1024)
o = torch.empty(M, 4096, dtype=torch.bfloat16, device="cuda")
side = torch.cuda.Stream()


def timed(tile, n=8):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): K.gemm(L.GEMM_NT, x, w, out=o, tile=tile)
    e.record(); e.synchronize()
    return s.elapsed_time(e) / n * 1e3


for _ in range(3): timed(2, 3); timed(7, 3)
print(f"alone:            per-tile {timed(2):7.1f} us   persistent {timed(7):7.1f} us", flush=True)
for cus in (8, 16, 32):
    side_m = 128 * 2 * cus // 8  # tiles = (side_m / 128) * 8 column tiles = 2 * cus workgroups = cus CUs
    Kocc = 1 << 20
    a, b = r(side_m, Kocc), r(1024, Kocc)
    oo = torch.empty(side_m, 1024, dtype=torch.bfloat16, device="cuda")
    torch.cuda.synchronize()
    res = {}
    for tile in (2, 7):
        with torch.cuda.stream(side):
            K.gemm(L.GEMM_NT, a, b, out=oo, tile=1, allow_split_k=False)  # tens of milliseconds
        torch.cuda.current_stream().wait_stream  # (no wait: the measurement runs BESIDE it)
        import time; time.sleep(0.002)
        res[tile] = timed(tile)
        torch.cuda.synchronize()
    print(f"{cus:3d} CUs held:     per-tile {res[2]:7.1f} us   persistent {res[7]:7.1f} us", flush=True)
    del a, b, oo
