"""The persistent NT kernel (tile hint 7) against the per-tile kernel (hint 2): bit-identity on ragged and step shapes for every epilogue it carries (plain, residual,
SwiGLU forward, SwiGLU backward), then time on the step's NT shapes.  GPU box only."""
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


def same(name, ref, got):
    refs, gots = (ref, got) if isinstance(ref, tuple) else ((ref,), (got,))
    ok = all(torch.equal(a, b) for a, b in zip(refs, gots))
    print(f"bit-identity {name}: {ok}" + ("" if ok else "  " + " / ".join(f"max abs diff {(a.float() - b.float()).abs().max().item():.4g}, mismatches {(a != b).sum().item()}" for a, b in zip(refs, gots))), flush=True)
    assert ok, name


for (m, n, k) in ((256, 256, 128), (1000, 512, 128), (257, 264, 192), (4099, 1024, 4096), (M, 4096, 1024), (70000, 1024, 2048)):
    x, w = r(m, k), r(n, k)
    ref = K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False)
    got = torch.full_like(ref, 7.0)
    K.gemm(L.GEMM_NT, x, w, out=got, tile=7)
    same(f"plain {m} x {n} x {k}", ref, got)
    res = r(m, n)
    same(f"residual {m} x {n} x {k}", K.gemm(L.GEMM_NT, x, w, residual=res, tile=2, allow_split_k=False), K.gemm(L.GEMM_NT, x, w, residual=res, tile=7))
for (m, f, k) in ((1000, 256, 128), (4099, 1056, 256), (M, 3072, 1024)):
    x, wgu = r(m, k), r(2 * f, k)
    same(f"SwiGLU forward {m} x 2*{f} x {k}", K.gemm_gateup_swiglu(x, wgu, tile=2), K.gemm_gateup_swiglu(x, wgu, tile=7))
    dy, w2, gu = r(m, k), r(k, f), r(m, 2 * f)
    same(f"SwiGLU backward {m} x {f} x {k}", K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=2), K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=7))
x1 = r(M, 1024)
cases = [("QKV fwd N=4096 K=1024", lambda t: (lambda a=x1, b=r(4096, 1024), o=r(M, 4096): K.gemm(L.GEMM_NT, a, b, out=o, tile=t)), 2.0 * M * 4096 * 1024),
         ("dctx dgrad N=2048 K=1024", lambda t: (lambda a=x1, b=r(2048, 1024), o=r(M, 2048): K.gemm(L.GEMM_NT, a, b, out=o, tile=t)), 2.0 * M * 2048 * 1024),
         ("dqkv dgrad N=1024 K=4096", lambda t: (lambda a=r(M, 4096), b=r(1024, 4096), o=r(M, 1024): K.gemm(L.GEMM_NT, a, b, out=o, tile=t)), 2.0 * M * 1024 * 4096),
         ("out-proj + residual N=1024 K=2048", lambda t: (lambda a=r(M, 2048), b=r(1024, 2048), o=r(M, 1024): K.gemm(L.GEMM_NT, a, b, out=o, residual=x1, tile=t)), 2.0 * M * 1024 * 2048),
         ("down + residual N=1024 K=3072", lambda t: (lambda a=r(M, 3072), b=r(1024, 3072), o=r(M, 1024): K.gemm(L.GEMM_NT, a, b, out=o, residual=x1, tile=t)), 2.0 * M * 1024 * 3072),
         ("gate-up + SwiGLU fwd N=6144 K=1024", lambda t: (lambda a=x1, b=r(6144, 1024): K.gemm_gateup_swiglu(a, b, tile=t)), 2.0 * M * 6144 * 1024),
         ("down dgrad + SwiGLU bwd N=3072 K=1024", lambda t: (lambda a=x1, b=r(1024, 3072), gu=r(M, 6144): K.gemm_dgrad_swiglu_bwd(a, b, gu, tile=t)), 2.0 * M * 3072 * 1024)]
for name, mk, flop in cases:
    fns = {t: mk(t) for t in (2, 7)}
    res = []
    for rep in range(2):
        for tile in (2, 7):
            t = timed(fns[tile])
            res.append(f"tile {tile}: {t:7.1f} us {flop / t / 1e6:6.0f} TF")
    print(f"{name:38s} " + " | ".join(res), flush=True)
    del fns
