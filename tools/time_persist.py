"""The persistent NT kernel (tile hint 7) against the per-tile kernel (hint 2): bit-identity on ragged and step shapes, then time on the step's plain NT shapes.  GPU box only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 113440


def timed(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for (m, n, k) in ((256, 256, 128), (1000, 512, 128), (257, 264, 192), (4099, 1024, 4096), (M, 4096, 1024), (70000, 1024, 2048)):
    x, w = r(m, k), r(n, k)
    ref = K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False)
    got = torch.full_like(ref, 7.0)
    K.gemm(L.GEMM_NT, x, w, out=got, tile=7)
    torch.cuda.synchronize()
    same = torch.equal(ref, got)
    print(f"bit-identity {m} x {n} x {k}: {same}" + ("" if same else f"  max abs diff {(ref.float() - got.float()).abs().max().item():.4g}, mismatches {(ref != got).sum().item()}"), flush=True)
    assert same
x1 = r(M, 1024)
cases = [("QKV fwd N=4096 K=1024", x1, r(4096, 1024)), ("dctx dgrad N=2048 K=1024", x1, r(2048, 1024)), ("N=1024 K=1024", x1, r(1024, 1024)),
         ("dqkv dgrad N=1024 K=4096", r(M, 4096), r(1024, 4096)), ("N=1024 K=3072", r(M, 3072), r(1024, 3072)), ("N=3072 K=1024", x1, r(3072, 1024))]
for name, a, b in cases:
    o = K.gemm(L.GEMM_NT, a, b)
    flop = 2.0 * a.shape[0] * b.shape[0] * a.shape[1]
    res = []
    for rep in range(2):
        for tile in (2, 7):
            t = timed(lambda: K.gemm(L.GEMM_NT, a, b, out=o, tile=tile))
            res.append(f"tile {tile}: {t:7.1f} us {flop / t / 1e6:6.0f} TF")
    print(f"{name:28s} " + " | ".join(res), flush=True)
