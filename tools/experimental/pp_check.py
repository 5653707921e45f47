"""Ping-pong persistent NT kernel (tile hint 8) against the persistent (7) and per-tile (2) kernels: correctness on ragged shapes, then time at the step's shapes.
    python tools/experimental/pp_check.py [check|time|both] [batch]
Integer operands make every product and sum exact, so the rotated K order of hint 8 must give the same BITS as hint 2; random operands: within fp32 summation noise."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K

BF16 = torch.bfloat16
mode = sys.argv[1] if len(sys.argv) > 1 else "both"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 160
dev = "cuda"


def ints(*s, lo=-2, hi=3, g=None):
    return torch.randint(lo, hi, s, generator=g, device="cpu").to(torch.float32).to(BF16).to(dev)


def rnd(*s, g=None, scale=0.3):
    return (scale * torch.randn(*s, generator=g)).to(BF16).to(dev)


def ulps(a, b):
    """(worst |a - b| in units of max(one bf16 ulp of |b|, 1e-4 of b's rms) -- a value near zero that flips by fp32 summation noise is no error --, fraction of elements that differ)"""
    af, bf = a.float(), b.float()
    d = (af - bf).abs()
    unit = torch.maximum(bf.abs() * 2.0**-7, 1e-4 * bf.pow(2).mean().sqrt())
    return float((d / unit).max()), float((d > 0).float().mean())


def check():
    shapes = [(709 * 5 + 3, 1024, 1024), (2600, 4096, 512), (4099, 264, 384), (513, 776, 2048), (4200, 512, 320), (300, 256, 320), (256, 256, 320), (2049, 1288, 640), (5000, 3072, 1024)]
    bad = 0
    for M, N, Kd in shapes:
        g = torch.Generator().manual_seed(M + 3 * N + 7 * Kd)
        for data in ("int", "rand"):
            mk = (lambda *s: ints(*s, g=g)) if data == "int" else (lambda *s: rnd(*s, g=g))
            x, w, res = mk(M, Kd), mk(N, Kd), mk(M, N)
            ref = K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False)
            got = torch.full_like(ref, 7.0)
            K.gemm(L.GEMM_NT, x, w, out=got, tile=8)
            ref_r = K.gemm(L.GEMM_NT, x, w, residual=res, tile=2, allow_split_k=False)
            got_r = K.gemm(L.GEMM_NT, x, w, residual=res, tile=8)
            F = (N // 64) * 32
            wgu = mk(2 * F, Kd)
            gu2, a2 = K.gemm_gateup_swiglu(x, wgu, tile=2)
            gu8, a8 = K.gemm_gateup_swiglu(x, wgu, tile=8)
            dy, w2, gu = mk(M, Kd), mk(Kd, N), (rnd(M, 2 * N, g=g))
            b2 = K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=2)
            b8 = K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=8)
            torch.cuda.synchronize()
            rows = [("plain", got, ref), ("res", got_r, ref_r), ("swiglu_fwd.gu", gu8, gu2), ("swiglu_fwd.a", a8, a2), ("swiglu_bwd", b8, b2)]
            for name, a, b in rows:
                mu, frac = ulps(a, b)
                exact = torch.equal(a, b)
                ok = exact if data == "int" else (mu <= 1.01 and frac < 1e-3)
                if data == "rand" and name in ("swiglu_fwd.a", "swiglu_bwd"):
                    ok = mu <= 8 and frac < 1e-3  # a 1-ulp difference of the rounded projection moves the activation's last bits too
                print(f"{'ok ' if ok else 'BAD'} {M:6d} {N:5d} {Kd:5d} {data:4s} {name:14s} exact={exact} max_units={mu:.2f} differ={frac:.4%}", flush=True)
                bad += not ok
    # attention delta through the library's automatic choice
    os.environ["MI355_GEMM_PERSIST_MIN_TILES"] = "1"
    for B, S, Hq, D, d in ((6, 709, 16, 128, 1024), (14, 300, 4, 128, 512)):
        g = torch.Generator().manual_seed(B * S)
        for data in ("int", "rand"):
            mk = (lambda *s: ints(*s, g=g)) if data == "int" else (lambda *s: rnd(*s, g=g))
            dy, w, ctx = mk(B * S, d), mk(d, Hq * D), mk(B * S, Hq * D)
            lse = torch.randn(B, Hq, S, generator=g).to(dev)
            os.environ["MI355_GEMM_PP"] = "0"
            r0 = K.dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D)
            os.environ["MI355_GEMM_PP"] = "16"
            r1 = K.dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D)
            torch.cuda.synchronize()
            r0 = r0 if isinstance(r0, (tuple, list)) else (r0,)
            r1 = r1 if isinstance(r1, (tuple, list)) else (r1,)
            for i, (a, b) in enumerate(zip(r0, r1)):
                if not torch.is_tensor(a):
                    continue
                if a.dtype == BF16:
                    mu, frac = ulps(a, b)
                    ok = torch.equal(a, b) if data == "int" else (mu <= 1.01 and frac < 1e-3)
                    print(f"{'ok ' if ok else 'BAD'} delta B{B} S{S} {data} out[{i}] bf16 exact={torch.equal(a, b)} max_units={mu:.2f} differ={frac:.4%}", flush=True)
                else:
                    err = float((a.float() - b.float()).abs().max() / (b.float().abs().max() + 1e-9))
                    ok = torch.equal(a, b) if data == "int" else err < 2e-2
                    print(f"{'ok ' if ok else 'BAD'} delta B{B} S{S} {data} out[{i}] {a.dtype} shape {tuple(a.shape)} exact={torch.equal(a, b)} rel_max={err:.2e}", flush=True)
                bad += not ok
    os.environ.pop("MI355_GEMM_PP", None)
    os.environ.pop("MI355_GEMM_PERSIST_MIN_TILES", None)
    print("CHECK", "FAILED" if bad else "PASSED", bad, flush=True)
    return bad


def check_walk():
    """hint 7 with the weight-stationary walk == hint 2, bit for bit (same K order), every epilogue, ragged shapes"""
    bad = 0
    W = 7 + 2048
    for M, N, Kd in [(709 * 5 + 3, 1024, 1024), (2600, 4096, 512), (4099, 264, 384), (513, 776, 2048), (300, 256, 320), (2049, 1288, 640), (5000, 3072, 1024), (70000, 1536, 128)]:
        g = torch.Generator().manual_seed(M + N + Kd)
        x, w, res = rnd(M, Kd, g=g), rnd(N, Kd, g=g), rnd(M, N, g=g)
        ok = torch.equal(K.gemm(L.GEMM_NT, x, w, tile=W), K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False))
        ok &= torch.equal(K.gemm(L.GEMM_NT, x, w, residual=res, tile=W), K.gemm(L.GEMM_NT, x, w, residual=res, tile=2, allow_split_k=False))
        F = (N // 64) * 32
        wgu = rnd(2 * F, Kd, g=g)
        a, b = K.gemm_gateup_swiglu(x, wgu, tile=W), K.gemm_gateup_swiglu(x, wgu, tile=2)
        ok &= torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        dy, w2, gu = rnd(M, Kd, g=g), rnd(Kd, N, g=g), rnd(M, 2 * N, g=g)
        ok &= torch.equal(K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=W), K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=2))
        print("ok " if ok else "BAD", "walk", M, N, Kd, flush=True)
        bad += not ok
    os.environ["MI355_GEMM_PERSIST_MIN_TILES"] = "1"
    for B, S, Hq, D, d in ((6, 709, 16, 128, 1024), (14, 300, 4, 128, 512)):
        g = torch.Generator().manual_seed(B * S)
        dy, w, ctx = rnd(B * S, d, g=g), rnd(d, Hq * D, g=g), rnd(B * S, Hq * D, g=g)
        lse = torch.randn(B, Hq, S, generator=g).to(dev)
        os.environ["MI355_GEMM_WALK"] = "0"
        r0 = K.dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D)
        os.environ["MI355_GEMM_WALK"] = "1"
        r1 = K.dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D)
        ok = torch.equal(r0[0], r1[0]) and torch.equal(r0[1], r1[1])
        print("ok " if ok else "BAD", "walk delta", B, S, flush=True)
        bad += not ok
    os.environ.pop("MI355_GEMM_WALK", None)
    os.environ.pop("MI355_GEMM_PERSIST_MIN_TILES", None)
    print("WALK CHECK", "FAILED" if bad else "PASSED", flush=True)
    return bad


def timeit(fn, reps):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(reps + 1)]
    ev[0].record()
    for i in range(reps):
        fn()
        ev[i + 1].record()
    torch.cuda.synchronize()
    ts = sorted(ev[i].elapsed_time(ev[i + 1]) * 1e3 for i in range(reps))
    return ts[len(ts) // 2], ts[0]


def bench():
    M = batch * 709
    Mh = batch * 512
    g = torch.Generator().manual_seed(5)
    cases = []
    x1 = rnd(M, 1024, g=g, scale=1.0)
    cases.append(("qkv plain N4096 K1024", M, 4096, 1024, lambda t, w=rnd(4096, 1024, g=g, scale=0.03): K.gemm(L.GEMM_NT, x1, w, tile=t)))
    res = rnd(M, 1024, g=g, scale=1.0)
    x2 = rnd(M, 2048, g=g, scale=1.0)
    cases.append(("outproj +res N1024 K2048", M, 1024, 2048, lambda t, w=rnd(1024, 2048, g=g, scale=0.03): K.gemm(L.GEMM_NT, x2, w, residual=res, tile=t)))
    x3 = rnd(M, 3072, g=g, scale=1.0)
    cases.append(("down +res N1024 K3072", M, 1024, 3072, lambda t, w=rnd(1024, 3072, g=g, scale=0.03): K.gemm(L.GEMM_NT, x3, w, residual=res, tile=t)))
    cases.append(("gateup swiglu N6144 K1024", M, 6144, 1024, lambda t, w=rnd(6144, 1024, g=g, scale=0.03): K.gemm_gateup_swiglu(x1, w, tile=t)))
    gu = rnd(M, 6144, g=g, scale=1.0)
    cases.append(("down dgrad swiglu_bwd N3072 K1024", M, 3072, 1024, lambda t, w=rnd(1024, 3072, g=g, scale=0.03): K.gemm_dgrad_swiglu_bwd(x1, w, gu, tile=t)))
    x4 = rnd(M, 4096, g=g, scale=1.0)
    cases.append(("qkv dgrad N1024 K4096", M, 1024, 4096, lambda t, w=rnd(1024, 4096, g=g, scale=0.03): K.gemm(L.GEMM_NT, x4, w, tile=t)))
    cases.append(("gateup dgrad N1024 K6144", M, 1024, 6144, lambda t, w=rnd(1024, 6144, g=g, scale=0.03): K.gemm(L.GEMM_NT, gu, w, tile=t)))
    cases.append(("outproj dgrad plain N2048 K1024", M, 2048, 1024, lambda t, w=rnd(2048, 1024, g=g, scale=0.03): K.gemm(L.GEMM_NT, x1, w, tile=t)))
    if batch >= 32:
        xh = rnd(Mh, 1024, g=g, scale=1.0)
        wv = rnd(151936, 1024, g=g, scale=0.03)
        logits = torch.empty(Mh, 151936, dtype=BF16, device=dev)
        cases.append(("lm head fwd N151936 K1024", Mh, 151936, 1024, lambda t: K.gemm(L.GEMM_NT, xh, wv, out=logits, tile=t)))
        wvt = K.transpose(wv)
        dxh = torch.empty(Mh, 1024, dtype=BF16, device=dev)
        logits.normal_()
        cases.append(("lm head dgrad N1024 K151936", Mh, 1024, 151936, lambda t: K.gemm(L.GEMM_NT, logits, wvt, out=dxh, tile=t)))
    print(f"batch {batch}: M = {M}")
    for name, m, n, k, fn in cases:
        flop = 2.0 * m * n * k
        hints = (7, 8, 8 + 256)
        for t in hints:
            fn(t)
        torch.cuda.synchronize()
        out = {}
        for rnd_ in range(2):
            for t in hints:
                med, mn = timeit(lambda: fn(t), 6 if n < 100000 and k < 100000 else 3)
                out.setdefault(t, []).append(med)
        a, b, c = min(out[hints[0]]), min(out[hints[1]]), min(out[hints[2]])
        print(f"{name:36s} hint7 {a:9.1f} us ({flop / a / 1e9:6.1f} TF/s)   hint8 {b:9.1f} us ({flop / b / 1e9:6.1f} TF/s) {100 * (b - a) / a:+.1f} %   hint8-noepi {c:9.1f} us {100 * (c - a) / a:+.1f} %", flush=True)


if __name__ == "__main__":
    rc = 0
    if mode in ("check", "both"):
        rc = check_walk() + check()
    if mode in ("time", "both") and not rc:
        bench()
    sys.exit(1 if rc else 0)
