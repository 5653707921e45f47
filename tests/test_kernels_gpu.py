"""Kernel-level parity on a real MI355X: every C-ABI entry point vs the CPU oracle on the same seeded inputs.

Bit-exact for index/gather/copy work; floating-point tolerances are written next to each check.
"""

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from llm_quest_amd import kernels

    return kernels


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def dev(t):
    return t.cuda()


def assert_bf16_within_1ulp(got, ref, max_frac=2e-3, what=""):
    """bf16 results of fp32 math may differ by one rounding step when the fp32 reduction order differs; never more."""
    got, ref = got.cpu(), ref.cpu()
    assert got.shape == ref.shape and got.dtype == BF16 and ref.dtype == BF16

    def ordered(t):
        i = t.contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)

    d = (ordered(got) - ordered(ref)).abs()
    assert int(d.max()) <= 1, f"{what}: max bf16 ulp distance {int(d.max())}"
    frac = float((d != 0).float().mean())
    assert frac <= max_frac, f"{what}: {frac:.2e} of elements differ by 1 ulp (allowed {max_frac:.0e})"


# ----------------------------------------------------------------------------------------------- GEMM
GEMM_SHAPES = [
    (128, 128, 64), (256, 384, 128), (100, 256, 192), (1, 128, 64), (333, 104, 72), (1418, 1024, 1024), (709, 4096, 1024),
    (130, 100, 768), (257, 3072, 768),
]


@pytest.mark.parametrize("tile", [1, 2, 3, 4, 5])
@pytest.mark.parametrize("form", ["NT", "NN", "TN"])
@pytest.mark.parametrize("M,N,K_", GEMM_SHAPES)
def test_gemm_forms(K, form, M, N, K_, tile):
    from llm_quest_amd import _lib as L

    if form == "NN" and N % 8:
        pytest.skip("NN needs N % 8 == 0")
    if form == "TN" and (M % 8 or N % 8):
        pytest.skip("TN needs M,N % 8 == 0")
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K_)
    a = torch.randn(M, K_, generator=g).to(BF16)
    b = torch.randn(N, K_, generator=g).to(BF16)
    ref = a.float() @ b.float().t()
    if form == "NT":
        out = K.gemm(L.GEMM_NT, dev(a), dev(b), out_dtype=F32, tile=tile)
    elif form == "NN":
        out = K.gemm(L.GEMM_NN, dev(a), dev(b.t().contiguous()), out_dtype=F32, tile=tile)
    else:
        out = K.gemm(L.GEMM_TN, dev(a.t().contiguous()), dev(b.t().contiguous()), out_dtype=F32, tile=tile)
    # fp32 accumulation of exact bf16 products: only summation order differs
    err = rel_l2(out, ref)
    assert err < 2e-6, f"{form} {M}x{N}x{K_} tile {tile}: rel l2 {err}"


@pytest.mark.parametrize("form", ["NT", "NN", "TN"])
def test_gemm_single_barrier_loop_matches_two_barrier_loop(K, form):
    """Tile 4 (one barrier per phase, 5-stage ring) against tile 3, bit for bit (same MFMA, same K order), over K-tile counts
    1..6 and a long K, ragged M / N, repeated to give a hazard a chance to show."""
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(5)
    f = {"NT": L.GEMM_NT, "NN": L.GEMM_NN, "TN": L.GEMM_TN}[form]
    for (M, N, Kd) in [(304, 520, 32), (256, 256, 40), (520, 264, 96), (264, 776, 136), (512, 512, 160), (520, 520, 192), (776, 1032, 4096)]:
        if form == "NT":
            a, b = torch.randn(M, Kd, generator=g), torch.randn(N, Kd, generator=g)
        elif form == "NN":
            a, b = torch.randn(M, Kd, generator=g), torch.randn(Kd, N, generator=g)
        else:
            a, b = torch.randn(Kd, M, generator=g), torch.randn(Kd, N, generator=g)
        a, b = dev(a.to(BF16)), dev(b.to(BF16))
        ref = K.gemm(f, a, b, out_dtype=F32, allow_split_k=False, tile=3)
        for _ in range(5):
            got = K.gemm(f, a, b, out_dtype=F32, allow_split_k=False, tile=4)
            assert torch.equal(got, ref), (form, M, N, Kd)


def test_gemm_epilogues(K):
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(5)
    M, N, K_ = 197, 3072, 768
    a = torch.randn(M, K_, generator=g).to(BF16)
    w = (torch.randn(N, K_, generator=g) * 0.05).to(BF16)
    bias = torch.randn(N, generator=g)
    res32 = torch.randn(M, N, generator=g)
    acc = a.float() @ w.float().t()
    # bias + exact GELU, bf16 out
    out = K.gemm(L.GEMM_NT, dev(a), dev(w), out_dtype=BF16, bias=dev(bias), gelu=True)
    ref = torch.nn.functional.gelu(acc + bias).to(BF16)
    assert rel_l2(out, ref) < 3e-3  # one bf16 rounding of the output
    # bias + fp32 residual, fp32 out
    out = K.gemm(L.GEMM_NT, dev(a), dev(w), out_dtype=F32, bias=dev(bias), residual=dev(res32))
    assert rel_l2(out, acc + bias + res32) < 2e-6
    # bf16 residual / in-place accumulate
    c = dev(res32.to(BF16))
    K.gemm(L.GEMM_NT, dev(a), dev(w), out=c, residual=c)
    assert rel_l2(c, (acc + res32.to(BF16).float()).to(BF16)) < 3e-3
    # strided operands (views into wider buffers), as the fused QKV layout uses
    wide = dev(torch.randn(M, 2 * K_, generator=g).to(BF16))
    out = K.gemm(L.GEMM_NT, wide[:, K_:], dev(w), out_dtype=F32)
    assert rel_l2(out, wide[:, K_:].float().cpu() @ w.float().t()) < 2e-6


def test_gemm_split_k_weight_gradient_shapes(K):
    """Few output tiles + long K (weight gradients): the K-split path must agree with the unsplit kernel and fp32 math,
    including accumulation into an existing gradient."""
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(77)
    tokens, n_out, k_in = 5000, 256, 384
    dy = torch.randn(tokens, n_out, generator=g).to(BF16)
    x = torch.randn(tokens, k_in, generator=g).to(BF16)
    ref = dy.float().t() @ x.float()
    split = K.gemm(L.GEMM_TN, dev(dy), dev(x), out_dtype=F32)
    plain = K.gemm(L.GEMM_TN, dev(dy), dev(x), out_dtype=F32, allow_split_k=False)
    assert rel_l2(split, ref) < 2e-6 and rel_l2(plain, ref) < 2e-6
    grad = dev(torch.randn(n_out, k_in, generator=g).to(BF16))
    before = grad.float().cpu()
    K.gemm(L.GEMM_TN, dev(dy), dev(x), out=grad, residual=grad)
    assert rel_l2(grad, (before + ref).to(BF16)) < 3e-3
    # NT with a long K and a single tile also splits
    a = torch.randn(100, 8192, generator=g).to(BF16)
    b = torch.randn(64, 8192, generator=g).to(BF16)
    assert rel_l2(K.gemm(L.GEMM_NT, dev(a), dev(b), out_dtype=F32), a.float() @ b.float().t()) < 2e-6


@pytest.mark.parametrize("tile", [0, 1, 3, 4])
def test_gemm_grouped_matches_single_launches(K, tile):
    """One grouped launch == the same problems launched one by one (bit-exact: same tile kernel, same K order), for ragged
    shapes, strided operand views, accumulation into an existing gradient, bf16 and fp32 outputs."""
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(123)
    tokens = 1416
    dy_all = dev(torch.randn(tokens, 904, generator=g).to(BF16))
    xs = [dev(torch.randn(tokens, k, generator=g).to(BF16)) for k in (264, 520, 72)]
    views = [dy_all[:, :392], dy_all[:, 392:648], dy_all[:, 648:]]  # strided A operands (lda = 904)
    for odt, tol in ((F32, 2e-6), (BF16, 3e-3)):
        outs = [torch.empty((v.shape[1], x.shape[1]), dtype=odt, device="cuda") for v, x in zip(views, xs)]
        acc = dev(torch.randn(views[1].shape[1], xs[1].shape[1], generator=g)).to(odt)
        outs[1].copy_(acc)
        problems = [(v, x, o, o if i == 1 else None) for i, (v, x, o) in enumerate(zip(views, xs, outs))]
        K.gemm_grouped(L.GEMM_TN, problems, tile=tile)
        for i, (v, x, o) in enumerate(zip(views, xs, outs)):
            ref = v.float().t() @ x.float() + (acc.float() if i == 1 else 0)
            assert rel_l2(o, ref) < tol, (i, odt)
            if tile:
                single = K.gemm(L.GEMM_TN, v, x, out_dtype=odt, residual=acc if i == 1 else None, allow_split_k=False, tile=tile)
                assert torch.equal(single, o), (i, odt)
    # NT / NN forms through the same path
    a = dev(torch.randn(300, 160, generator=g).to(BF16))
    b = dev(torch.randn(520, 160, generator=g).to(BF16))
    o1 = torch.empty((300, 520), dtype=BF16, device="cuda")
    o2 = torch.empty((520, 300), dtype=BF16, device="cuda")
    K.gemm_grouped(L.GEMM_NT, [(a, b, o1, None), (b, a, o2, None)], tile=tile)
    assert rel_l2(o1, a.float() @ b.float().t()) < 3e-3 and rel_l2(o2, b.float() @ a.float().t()) < 3e-3
    bt = b.t().contiguous()
    o3 = torch.empty((300, 520), dtype=F32, device="cuda")
    K.gemm_grouped(L.GEMM_NN, [(a, bt, o3, None)], tile=tile)
    assert rel_l2(o3, a.float() @ b.float().t()) < 2e-6
    with pytest.raises(RuntimeError):
        K.gemm_grouped(L.GEMM_TN, [(dy_all, xs[0], torch.empty((904, 264), dtype=F32, device="cuda"), None)], tile=2)


@pytest.mark.gpu
def test_transpose_and_nt_dgrad_match_nn_form(K):
    """mi355_transpose_bf16 is a bit-exact transpose (ragged tile edges, strided views), and the dgrad it enables -- dX = dY W as an NT GEMM on
    W^T -- returns the NN form's bits, plain and with the SwiGLU backward in the epilogue."""
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(11)
    for R, C in [(64, 64), (1024, 3072), (200, 136), (8, 1032), (6144, 1024)]:
        x = torch.randn(R, C + 8, generator=g).to(BF16)
        xv = dev(x)[:, :C]  # row-strided view
        y = K.transpose(xv)
        assert y.shape == (C, R) and torch.equal(y.cpu(), x[:, :C].t().contiguous()), (R, C)
    with pytest.raises(RuntimeError):
        K.transpose(dev(torch.zeros(12, 64).to(BF16)))  # rows not a multiple of 8: refused, not mis-copied
    M, d, F_ = 4352, 256, 640
    dy = dev(torch.randn(M, d, generator=g).to(BF16))
    w = dev((torch.randn(d, F_, generator=g) * 0.1).to(BF16))
    nn = K.gemm(L.GEMM_NN, dy, w)
    assert K.DGRAD_NT and M >= K.DGRAD_NT_MIN_ROWS
    assert torch.equal(K.dgrad(dy, w), nn)
    small = dy[:512]
    assert torch.equal(K.dgrad(small, w), K.gemm(L.GEMM_NN, small, w))  # below the threshold: NN form, no transpose
    gu = dev(torch.randn(M, 2 * F_, generator=g).to(BF16))
    fused = K.gemm_dgrad_swiglu_bwd(dy, w, gu)
    two_step = K.swiglu_bwd(gu, nn, F_)
    assert torch.equal(fused, two_step)


@pytest.mark.parametrize("M,N,dtype", [(1500, 200, BF16), (50432, 768, BF16), (4100, 3075, BF16), (777, 1032, F32), (33, 8, F32)])
def test_colsum(K, M, N, dtype):
    """Bias-gradient column sums (mi355_colsum): 16-byte loads on aligned column groups, the scalar tail on the last columns of a width that is no
    multiple of 8, a row-strided view (the QKV bias reads a slice of a wider matrix), accumulation into an existing gradient."""
    g = torch.Generator().manual_seed(9)
    x = torch.randn(M, N, generator=g).to(dtype)
    want = x.double().sum(0)
    out = K.colsum(dev(x))
    assert rel_l2(out, want) < 2e-6 * math.sqrt(M) + 1e-6
    K.colsum(dev(x), out=out, accumulate=True)
    assert rel_l2(out, 2 * want) < 2e-6 * math.sqrt(M) + 1e-6
    if N >= 64:
        wide = dev(torch.cat([x, x], dim=1))
        view = wide[:, N // 2 : N // 2 + (N // 8) * 8 // 2]  # a column slice: pitch != width, start not on a 16-byte boundary when N // 2 is odd
        assert rel_l2(K.colsum(view), view.double().sum(0)) < 2e-6 * math.sqrt(M) + 1e-6


# ----------------------------------------------------------------------------------------------- norms / rope
def test_rmsnorm_matches_golden(K, golden):
    from oracle import ops

    t = golden("per_op")
    for width in (1024, 128):
        x, w = t[f"rmsnorm.{width}.x"], t[f"rmsnorm.{width}.w"]
        y, rstd = K.rmsnorm_fwd(dev(x), dev(w))
        assert torch.equal(y.cpu(), t[f"rmsnorm.{width}.y"]), f"rmsnorm fwd width {width} not bit-exact"
        dx, dw = K.rmsnorm_bwd(dev(x), dev(w), rstd, dev(t[f"rmsnorm.{width}.gy"]))
        # reference grads are bf16 autograd results; ours come from fp32 math rounded once
        assert rel_l2(dx, t[f"rmsnorm.{width}.gx"]) < 8e-3
        assert rel_l2(dw, t[f"rmsnorm.{width}.gw"]) < 8e-3
    # larger random case vs the oracle, with the residual-grad fusion
    g = torch.Generator().manual_seed(3)
    x = (torch.randn(777, 1024, generator=g) * 2).to(BF16)
    w = (1 + 0.1 * torch.randn(1024, generator=g)).to(BF16)
    dy = torch.randn(777, 1024, generator=g).to(BF16)
    dres = torch.randn(777, 1024, generator=g).to(BF16)
    y, rstd = K.rmsnorm_fwd(dev(x), dev(w))
    assert_bf16_within_1ulp(y, ops.rmsnorm(x, w), what="rmsnorm fwd 777x1024")
    xr = x.float().requires_grad_(True)
    wr = w.float().requires_grad_(True)
    inv = torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-6)
    (xr * inv * wr).backward(dy.float())
    dx, dw = K.rmsnorm_bwd(dev(x), dev(w), rstd, dev(dy), dev(dres))
    assert rel_l2(dx, xr.grad + dres.float()) < 4e-3  # bf16 output rounding
    assert rel_l2(dw, wr.grad) < 1e-4


def _qkv_case(tokens_b, S, Hq, Hkv, D, seed):
    g = torch.Generator().manual_seed(seed)
    qkv = torch.randn(tokens_b * S, (Hq + 2 * Hkv) * D, generator=g).to(BF16)
    qw = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    kw = (1 + 0.1 * torch.randn(D, generator=g)).to(BF16)
    return qkv, qw, kw


@pytest.mark.parametrize("D", [128, 64])
def test_qknorm_rope(K, D):
    from oracle import ops

    B, S, Hq, Hkv = 2, 50, 4, 2
    qkv, qw, kw = _qkv_case(B, S, Hq, Hkv, D, 11)
    cos, sin = ops.rope_tables(1_000_000, D, 96)
    pos = torch.arange(S, dtype=torch.int32).repeat(B)
    q, k, rstd = K.qknorm_rope_fwd(dev(qkv), dev(qw), dev(kw), dev(cos), dev(sin), dev(pos), Hq, Hkv, D)
    q4 = qkv[:, : Hq * D].view(B, S, Hq, D).transpose(1, 2)
    k4 = qkv[:, Hq * D : (Hq + Hkv) * D].view(B, S, Hkv, D).transpose(1, 2)
    q_ref = ops.rope_apply(ops.rmsnorm(q4, qw), cos, sin).transpose(1, 2).reshape(B * S, Hq * D)
    k_ref = ops.rope_apply(ops.rmsnorm(k4, kw), cos, sin).transpose(1, 2).reshape(B * S, Hkv * D)
    # same rounding points as the reference; only the fp32 sum-of-squares order can differ
    assert_bf16_within_1ulp(q, q_ref, what="fused QK-norm + RoPE (q)")
    assert_bf16_within_1ulp(k, k_ref, what="fused QK-norm + RoPE (k)")
    # backward vs fp32 autograd of the same math
    g = torch.Generator().manual_seed(12)
    dq = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    dk = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    x32 = qkv.float().requires_grad_(True)
    qw32, kw32 = qw.float().requires_grad_(True), kw.float().requires_grad_(True)
    cb, sb = cos[:S].to(BF16).float(), sin[:S].to(BF16).float()

    def path(x, w, H):
        x = x.view(B, S, H, D).transpose(1, 2)
        n = x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + 1e-6) * w
        rot = torch.cat((-n[..., D // 2 :], n[..., : D // 2]), -1)
        return (cb * n + sb * rot).transpose(1, 2).reshape(B * S, H * D)

    yq = path(x32[:, : Hq * D], qw32, Hq)
    yk = path(x32[:, Hq * D : (Hq + Hkv) * D], kw32, Hkv)
    (yq * dq.float()).sum().add((yk * dk.float()).sum()).backward()
    dqkv = torch.zeros_like(dev(qkv))
    dqw, dkw = K.qknorm_rope_bwd(dev(qkv), dev(qw), dev(kw), dev(cos), dev(sin), dev(pos), rstd, dev(dq), dev(dk), dqkv, Hq, Hkv, D)
    n_qk = (Hq + Hkv) * D
    assert rel_l2(dqkv[:, :n_qk], x32.grad[:, :n_qk]) < 4e-3
    assert torch.count_nonzero(dqkv[:, n_qk:]) == 0
    assert rel_l2(dqw, qw32.grad) < 1e-4 and rel_l2(dkw, kw32.grad) < 1e-4


def test_rope_position_ids(K, golden):
    """position_ids path of RoPE.apply (golden rope.y_pid) through the fused kernel with unit norm weights."""
    t = golden("per_op")
    x = t["rope.x"]  # (2,4,64,128) bf16
    B, H, S, D = x.shape
    # the fused kernel always normalises first, so the expected value is oracle RMSNorm (unit weight) -> oracle RoPE
    from oracle import ops

    w = torch.ones(D, dtype=BF16)
    normed = ops.rmsnorm(x, w)
    ref = ops.rope_apply(normed, t["rope.cos"], t["rope.sin"], t["rope.pid"])
    assert torch.equal(ops.rope_apply(x, t["rope.cos"], t["rope.sin"], t["rope.pid"]), t["rope.y_pid"])
    # token-major [q heads | one k head | one v head]; k/v are zeros
    qkv = torch.cat([x.transpose(1, 2).reshape(B * S, H * D), torch.zeros(B * S, 2 * D, dtype=BF16)], dim=1)
    q, k, _ = K.qknorm_rope_fwd(dev(qkv), dev(w), dev(w), dev(t["rope.cos"]), dev(t["rope.sin"]), dev(t["rope.pid"].reshape(-1).to(torch.int32)), H, 1, D)
    assert_bf16_within_1ulp(q, ref.transpose(1, 2).reshape(B * S, H * D).contiguous(), what="RoPE position_ids path")


def test_layernorm(K, golden):
    t = golden("per_op")
    y = K.layernorm_fwd(dev(t["layernorm.x"]), dev(t["layernorm.scale"]), dev(t["layernorm.shift"]), out_dtype=F32)
    assert rel_l2(y, t["layernorm.y"]) < 1e-6
    yb = K.layernorm_fwd(dev(t["layernorm.x"]), dev(t["layernorm.scale"]), dev(t["layernorm.shift"]), out_dtype=BF16)
    assert rel_l2(yb, t["layernorm.y"]) < 3e-3


@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("width", [768, 1024, 192])
def test_layernorm_backward_row_in_registers(K, mode, width):
    """LayerNorm backward (mode 0: the reference ViT's sigma + eps; mode 1: nn.LayerNorm) against autograd in fp64, for the
    register-resident kernel (widths 768 / 1024) and the generic one (192), bf16 and fp32 upstream gradients, fused residual."""
    g = torch.Generator().manual_seed(width + mode)
    rows = 333
    x = torch.randn(rows, width, generator=g) * 2 + 0.5
    sc, sh = 1 + 0.2 * torch.randn(width, generator=g), 0.1 * torch.randn(width, generator=g)
    dres = torch.randn(rows, width, generator=g)
    eps = 1e-5 if mode == 0 else 1e-6
    for dy in (torch.randn(rows, width, generator=g), torch.randn(rows, width, generator=g).to(BF16)):
        xd, scd, shd = x.double().requires_grad_(True), sc.double().requires_grad_(True), sh.double().requires_grad_(True)
        mu = xd.mean(-1, keepdim=True)
        var = ((xd - mu) ** 2).mean(-1, keepdim=True)
        y = scd * (xd - mu) / (var.sqrt() + eps) + shd if mode == 0 else scd * (xd - mu) / (var + eps).sqrt() + shd
        y.backward(dy.double())
        _, mean, rsig = K.layernorm_fwd(dev(x), dev(sc), dev(sh), out_dtype=F32, eps=eps, want_stats=True, mode=mode)
        dx, dsc, dsh = K.layernorm_bwd(dev(x), dev(sc), mean, rsig, dev(dy), dres=dev(dres), eps=eps, mode=mode)
        assert rel_l2(dx, xd.grad + dres.double()) < 2e-6
        assert rel_l2(dsc, scd.grad) < 2e-6 and rel_l2(dsh, shd.grad) < 2e-6


def test_swiglu(K, golden):
    g = torch.Generator().manual_seed(21)
    gu = torch.randn(300, 2 * 3072, generator=g).to(BF16)
    a = K.swiglu_fwd(dev(gu), 3072)
    u, gt = gu[:, :3072], gu[:, 3072:]
    ref = u * torch.nn.functional.silu(gt)  # bf16 tensors, same rounding points as the reference FFN
    # silu on bf16 CPU may round differently by 1 ulp from our fp32 expf path
    assert rel_l2(a, ref) < 2e-3
    assert (a.cpu().float() - ref.float()).abs().max() <= 2 * ref.float().abs().max() * 2 ** -8
    da = torch.randn(300, 3072, generator=g).to(BF16)
    dgu = K.swiglu_bwd(dev(gu), dev(da), 3072)
    u32, g32 = u.float().requires_grad_(True), gt.float().requires_grad_(True)
    (u32 * torch.nn.functional.silu(g32)).backward(da.float())
    assert rel_l2(dgu[:, :3072], u32.grad) < 4e-3 and rel_l2(dgu[:, 3072:], g32.grad) < 4e-3


# ----------------------------------------------------------------------------------------------- attention
def _attn_ref(q, k, v, B, S, Hq, Hkv, D, key_mask, causal):
    """fp64 reference with the reference's finite mask fill; returns o [B*S,Hq*D]."""
    q4 = q.double().view(B, S, Hq, D).transpose(1, 2)
    k4 = k.double().view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    v4 = v.double().view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    s = (q4 @ k4.mT) * D ** -0.5
    blocked = torch.zeros(B, 1, S, S, dtype=torch.bool, device=q.device)
    if causal:
        blocked = blocked | torch.triu(torch.ones(S, S, dtype=torch.bool, device=q.device), 1)
    if key_mask is not None:
        blocked = blocked | ~key_mask.bool().to(q.device)[:, None, None, :]
    s = s.masked_fill(blocked, float(torch.finfo(torch.bfloat16).min) / 2)
    p = torch.softmax(s, dim=-1)
    return (p @ v4).transpose(1, 2).reshape(B * S, Hq * D), p


ATTN_CASES = [
    # B, S, Hq, Hkv, D, causal, ragged
    (2, 709, 4, 2, 128, True, False),
    (2, 709, 4, 2, 128, True, True),
    (1, 64, 2, 1, 128, True, False),
    (2, 197, 3, 3, 64, False, False),
    (1, 130, 2, 2, 64, True, True),
    (1, 33, 2, 1, 128, False, False),
    (2, 200, 4, 2, 128, True, "holes"),   # every fifth key padded: mask words with zeros in both halves (a sign-extension bug hid here)
    (2, 200, 2, 2, 128, False, "holes"),
]


@pytest.mark.parametrize("B,S,Hq,Hkv,D,causal,ragged", ATTN_CASES)
def test_attention_fwd_bwd(K, B, S, Hq, Hkv, D, causal, ragged):
    g = torch.Generator().manual_seed(S + D + Hq)
    q = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    k = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    v = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    do = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    km = None
    if ragged:
        km = torch.ones(B, S, dtype=torch.uint8)
        if ragged == "holes":
            km[:, 2::5] = 0
        else:
            km[0, S - S // 3 :] = 0
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, B, S, Hq, Hkv, D, km, causal)
    o_ref.backward(do.double())
    o, lse = K.attn_fwd(dev(q), dev(k), dev(v), B, S, Hq, Hkv, D, key_mask=None if km is None else dev(km), causal=causal)
    # P is rounded to bf16 before PV and O is stored in bf16: ~2^-9 relative
    assert rel_l2(o, o_ref) < 4e-3, f"fwd rel l2 {rel_l2(o, o_ref)}"
    dq, dk, dv = (torch.zeros_like(dev(t)) for t in (q, k, v))
    K.attn_bwd(dev(q), dev(k), dev(v), o, dev(do), lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=None if km is None else dev(km), causal=causal)
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        e = rel_l2(got, ref)
        assert e < 8e-3, f"{name} rel l2 {e}"


@pytest.mark.parametrize("B,S,Hq,Hkv,causal,ragged", [(2, 709, 4, 2, True, True), (1, 333, 2, 2, False, False), (3, 64, 4, 1, True, False), (1, 130, 2, 1, True, True),
                                                      (2, 200, 4, 2, True, "holes")])
def test_attention_backward_scratch_form_equals_recompute_form(K, B, S, Hq, Hkv, causal, ragged):
    """mi355_attn_bwd_ws (the dK/dV pass leaves dS in a scratch buffer, dQ = scale * dS K is one product over it) against the form
    without a workspace (the dQ pass recomputes S and dP).  The two were bit-identical while they shared their arithmetic; since round 3 the
    scratch form's dK/dV pass folds the scale into its bf16 K rows and starts its chains from -lse / -delta (one more operand rounding, ~1e-3
    of a score), so they agree to the bf16 level of a gradient, and each is checked against fp64 elsewhere; a scratch buffer that is too small
    is refused."""
    D = 128
    g = torch.Generator().manual_seed(S + Hq)
    q, k, v, do = (dev(torch.randn(B * S, w * D, generator=g).to(BF16)) for w in (Hq, Hkv, Hkv, Hq))
    km = None
    if ragged:
        km = torch.ones(B, S, dtype=torch.uint8)
        if ragged == "holes":
            km[:, 2::5] = 0
        else:
            km[0, S - S // 3 :] = 0
        km = dev(km)
    o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=causal)
    outs = []
    for spill in (False, True):
        K._ATTN_DS_SPILL = spill
        try:
            dq, dk, dv = torch.full_like(q, float("nan")), torch.full_like(k, float("nan")), torch.full_like(v, float("nan"))
            K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=km, causal=causal)
            outs.append((dq, dk, dv))
        finally:
            K._ATTN_DS_SPILL = True
    for a, b in zip(*outs):
        assert torch.isfinite(b.float()).all() and rel_l2(a, b) < 6e-3
    from llm_quest_amd import _lib as L

    need = L.load().mi355_attn_bwd_workspace_bytes(B, S, Hq, D)
    rows = (B * Hq * S * 4 + 15) // 16 * 16  # -lse * log2(e) and -delta, fp32 [B, Hq, S] each, behind the dS scratch
    assert need == B * Hq * ((S + 127) // 128 * 128) ** 2 * 2 + 2 * rows and L.load().mi355_attn_bwd_workspace_bytes(B, S, Hq, 64) == 0
    ws = torch.empty(need - 16, dtype=torch.uint8, device="cuda")
    delta = torch.empty_like(lse)
    L.require_gpu(q, ws)
    with pytest.raises(RuntimeError, match="workspace"):
        L.call("mi355_attn_bwd_ws", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0), L.ptr(do), do.stride(0),
               L.ptr(lse), L.ptr(delta), L.ptr(dq), dq.stride(0), L.ptr(dk), dk.stride(0), L.ptr(dv), dv.stride(0), L.ptr(km), int(causal), D ** -0.5, L.ptr(ws), need - 16)


def test_attention_backward_scratch_is_capped_and_falls_back_to_the_recompute_form(K, monkeypatch):
    """The dS scratch grows with S^2; above the cap (MI355_ATTN_DS_SPILL_MAX_MB) -- or without memory for it -- ``attn_bwd`` must take the
    O(S)-memory recompute form instead of allocating: S = 4096, B = 4 with the cap lowered under the 1 GiB this shape asks for.  Checked against
    fp64 (evaluated on the GPU by torch: 1 GiB of scores per tensor) and against the scratch form, bit for bit."""
    B, S, Hq, Hkv, D = 4, 4096, 2, 1, 128
    g = torch.Generator().manual_seed(11)
    q, k, v, do = (dev(torch.randn(B * S, w * D, generator=g).to(BF16)) for w in (Hq, Hkv, Hkv, Hq))
    o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
    from llm_quest_amd import _lib as L

    need = L.load().mi355_attn_bwd_workspace_bytes(B, S, Hq, D)
    assert need == B * Hq * S * S * 2 + 2 * B * Hq * S * 4
    K.release_attention_scratch()
    monkeypatch.setattr(K, "_ATTN_DS_SPILL_MAX", need - 1)
    before = dict(K.attn_bwd_form)
    torch.cuda.synchronize()
    base = torch.cuda.memory_allocated()
    torch.cuda.reset_peak_memory_stats()
    dq, dk, dv = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, causal=True)
    torch.cuda.synchronize()
    assert K.attn_bwd_form["recompute"] == before["recompute"] + 1 and K.attn_bwd_form["spill"] == before["spill"]
    assert torch.cuda.max_memory_allocated() - base < need // 4 and not K._ATTN_WS  # nothing of the scratch's size was allocated or cached
    monkeypatch.setattr(K, "_ATTN_DS_SPILL_MAX", need)
    dq2, dk2, dv2 = torch.zeros_like(q), torch.zeros_like(k), torch.zeros_like(v)
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq2, dk2, dv2, causal=True)
    assert K.attn_bwd_form["spill"] == before["spill"] + 1 and len(K._ATTN_WS) == 1
    assert rel_l2(dq, dq2) < 6e-3 and rel_l2(dk, dk2) < 6e-3 and rel_l2(dv, dv2) < 6e-3  # (the two forms round one operand differently)
    # a much smaller request on the same stream replaces the big buffer instead of pinning it
    K.attn_bwd(q[: 2 * 128], k[: 2 * 128], v[: 2 * 128], o[: 2 * 128], do[: 2 * 128], lse[:2, :, :128].contiguous(), 2, 128, Hq, Hkv, D,
               dq2[: 2 * 128], dk2[: 2 * 128], dv2[: 2 * 128], causal=True)
    assert max(w.numel() for w in K._ATTN_WS.values()) < need // 4
    K.release_attention_scratch()
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    o_ref, _ = _attn_ref(qr, kr, vr, B, S, Hq, Hkv, D, None, True)
    o_ref.backward(do.double())
    assert rel_l2(o, o_ref) < 4e-3
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        assert rel_l2(got, ref) < 8e-3, name


@pytest.mark.parametrize("B,S,Hq,Hkv,ragged", [(2, 709, 4, 2, False), (3, 300, 2, 2, True), (1, 128, 2, 1, False)])
def test_attention_backward_with_the_query_norm_backward_as_its_write_out(K, B, S, Hq, Hkv, ragged):
    """mi355_attn_bwd_qnorm (the dQ pass ends in the QK-norm + RoPE backward of the query heads; dQ is never a matrix) against the pair it
    replaces, attn_bwd + qknorm_rope_bwd: the same d(qkv) and norm-weight gradient up to the bf16 rounding of dQ the pair goes through (the fused
    form is the more exact one: checked against fp32 autograd of the norm + RoPE fed with the pair's own dQ), the key heads' rows bit for bit,
    dK / dV bit for bit, the weight gradient bit-reproducible."""
    from oracle import ops

    D = 128
    qkv, qw, kw = _qkv_case(B, S, Hq, Hkv, D, 31)
    cos, sin = ops.rope_tables(1_000_000, D, 1024)
    pos = torch.arange(S, dtype=torch.int32).repeat(B)
    km = None
    if ragged:
        km = torch.ones(B, S, dtype=torch.uint8)
        km[0, S - 37 :] = 0
        km[1, S // 2 :] = 0
        km = dev(km)
    qkv_d, qw_d, kw_d, cos_d, sin_d, pos_d = dev(qkv), dev(qw), dev(kw), dev(cos), dev(sin), dev(pos)
    q, k, rstd = K.qknorm_rope_fwd(qkv_d, qw_d, kw_d, cos_d, sin_d, pos_d, Hq, Hkv, D)
    v = qkv_d[:, (Hq + Hkv) * D :]
    o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
    g = torch.Generator().manual_seed(32)
    do = dev(torch.randn(B * S, Hq * D, generator=g).to(BF16))
    # the pair
    dq, dk0, dqkv0 = torch.empty_like(q), torch.empty_like(k), torch.zeros_like(qkv_d)
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk0, dqkv0[:, (Hq + Hkv) * D :], key_mask=km, causal=True)
    dqw0, dkw0 = K.qknorm_rope_bwd(qkv_d, qw_d, kw_d, cos_d, sin_d, pos_d, rstd, dq, dk0, dqkv0, Hq, Hkv, D)
    # the fused form
    dk1, dqkv1 = torch.empty_like(k), torch.zeros_like(qkv_d)
    dqw1 = K.attn_bwd_qnorm(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dk1, dqkv1[:, (Hq + Hkv) * D :], qkv_d, qw_d, cos_d, sin_d, pos_d, rstd, dqkv1, key_mask=km, causal=True)
    assert dqw1 is not None, "the fused form must apply at head_dim 128 with the scratch available"
    z, dkw1 = K.qknorm_rope_bwd(qkv_d, qw_d, kw_d, cos_d, sin_d, pos_d, rstd, None, dk1, dqkv1, Hq, Hkv, D)
    assert torch.equal(dk0, dk1) and torch.equal(dqkv0[:, (Hq + Hkv) * D :], dqkv1[:, (Hq + Hkv) * D :]), "dK / dV do not depend on the dQ write-out"
    assert torch.equal(dqkv0[:, Hq * D : (Hq + Hkv) * D], dqkv1[:, Hq * D : (Hq + Hkv) * D]) and torch.equal(dkw0, dkw1), "key heads: the same kernel, the same bits"
    assert torch.count_nonzero(z) == 0
    nq = Hq * D
    assert rel_l2(dqkv1[:, :nq], dqkv0[:, :nq]) < 4e-3, rel_l2(dqkv1[:, :nq], dqkv0[:, :nq])
    assert rel_l2(dqw1, dqw0) < 2e-3, rel_l2(dqw1, dqw0)
    # fp32 autograd of the norm + RoPE fed with the pair's (bf16) dQ: both forms are within bf16 rounding of it, the fused one no further than the pair
    x32 = qkv[:, :nq].float().requires_grad_(True)
    w32 = qw.float().requires_grad_(True)
    cb, sb = cos[:S].to(BF16).float(), sin[:S].to(BF16).float()
    xh = x32.view(B, S, Hq, D).transpose(1, 2)
    n = xh * torch.rsqrt(xh.pow(2).mean(-1, keepdim=True) + 1e-6) * w32
    y = (cb * n + sb * torch.cat((-n[..., D // 2 :], n[..., : D // 2]), -1)).transpose(1, 2).reshape(B * S, nq)
    (y * dq.float().cpu()).sum().backward()
    e0, e1 = rel_l2(dqkv0[:, :nq], x32.grad), rel_l2(dqkv1[:, :nq], x32.grad)
    assert e0 < 4e-3 and e1 < 5e-3, (e0, e1)
    assert rel_l2(dqw1, w32.grad) < 3e-3 and rel_l2(dqw0, w32.grad) < 1e-4
    # rows of masked / out-of-range queries contribute nothing odd, and the result is bit-reproducible
    dk2, dqkv2 = torch.empty_like(k), torch.zeros_like(qkv_d)
    dqw2 = K.attn_bwd_qnorm(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dk2, dqkv2[:, (Hq + Hkv) * D :], qkv_d, qw_d, cos_d, sin_d, pos_d, rstd, dqkv2, key_mask=km, causal=True)
    assert torch.equal(dqw1, dqw2) and torch.equal(dqkv1[:, :nq], dqkv2[:, :nq])
    assert torch.isfinite(dqkv1.float()).all() and torch.isfinite(dqw1).all()
    # the compact bf16 coefficient table (plain RoPE: both halves of cos / sin equal) and the fp32 tables give the same bits
    assert K.rope_cs16(cos_d, sin_d) is not None
    keep, K.ROPE_CS16 = K.ROPE_CS16, False
    try:
        assert K.rope_cs16(cos_d, sin_d) is None
        dk3, dqkv3 = torch.empty_like(k), torch.zeros_like(qkv_d)
        dqw3 = K.attn_bwd_qnorm(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dk3, dqkv3[:, (Hq + Hkv) * D :], qkv_d, qw_d, cos_d, sin_d, pos_d, rstd, dqkv3, key_mask=km, causal=True)
    finally:
        K.ROPE_CS16 = keep
    assert torch.equal(dqw1, dqw3) and torch.equal(dqkv1[:, :nq], dqkv3[:, :nq])
    skew = cos_d.clone()
    skew[:, D // 2 :] += 1e-3
    assert K.rope_cs16(skew, sin_d) is None, "tables whose halves differ must take the fp32 path"


@pytest.mark.parametrize("persistent", [False, True])
@pytest.mark.parametrize("B,S,Hq,Hkv", [(6, 709, 4, 2), (9, 500, 3, 1)])
def test_out_projection_dgrad_leaves_the_attention_backwards_row_constants(K, B, S, Hq, Hkv, persistent, monkeypatch):
    """(``persistent``: the same epilogue in the persistent NT kernel, which the library takes by itself from 512 tiles upward -- forced here from one tile.)
    mi355_gemm_bf16_attn_delta: d(ctx) bit-identical to the plain dgrad; delta[b, h, s] = sum_d d(ctx) * ctx on the ROUNDED d(ctx) (what the
    stand-alone delta pass reads; fp32 sums in another order: 1e-5), -delta and -lse * log2(e) exact in the backward's scratch; an odd head count
    (the last 256-column tile holds one head) and a row count that is no multiple of 256; and the attention backward that skips its delta pass
    (MI355_ATTN_DELTA_READY) gives the gradients of the one that runs it, to the rounding of delta."""
    from llm_quest_amd import _lib as L
    from oracle import ops

    monkeypatch.setenv("MI355_GEMM_PERSIST_MIN_TILES", "1" if persistent else "1000000000")
    D, d_model = 128, 512
    qkv, qw, kw = _qkv_case(B, S, Hq, Hkv, D, 41)
    cos, sin = ops.rope_tables(1_000_000, D, 1024)
    pos = torch.arange(S, dtype=torch.int32).repeat(B)
    qkv_d, qw_d, kw_d, cos_d, sin_d, pos_d = dev(qkv), dev(qw), dev(kw), dev(cos), dev(sin), dev(pos)
    q, k, rstd = K.qknorm_rope_fwd(qkv_d, qw_d, kw_d, cos_d, sin_d, pos_d, Hq, Hkv, D)
    v = qkv_d[:, (Hq + Hkv) * D :]
    ctx, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
    g = torch.Generator().manual_seed(42)
    dy = dev((0.05 * torch.randn(B * S, d_model, generator=g)).to(BF16))
    w = dev((0.05 * torch.randn(d_model, Hq * D, generator=g)).to(BF16))  # out_proj.weight [d_model, Hq*D]
    plain = K.dgrad(dy, w)
    fused = K.dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D)
    assert fused is not None, "the fused form must apply at head_dim 128 with >= 4096 rows and the scratch available"
    dctx, delta = fused
    assert torch.equal(dctx, plain)
    want = (plain.float() * ctx.float()).view(B, S, Hq, D).sum(-1).permute(0, 2, 1).contiguous()
    assert rel_l2(delta, want) < 1e-5 and delta.shape == lse.shape
    lib = L.load()
    need = lib.mi355_attn_bwd_workspace_bytes(B, S, Hq, D)
    ws = K._attn_scratch(dy.device, need)
    n = B * Hq * S
    offs = [lib.mi355_attn_bwd_workspace_rowconst_offset(B, S, Hq, D, i) for i in (0, 1)]
    assert offs[0] > 0 and offs[1] >= offs[0] + 4 * n and offs[1] + 4 * n <= need
    nl2 = ws[offs[0] : offs[0] + 4 * n].view(torch.float32).view(B, Hq, S)
    ndl = ws[offs[1] : offs[1] + 4 * n].view(torch.float32).view(B, Hq, S)
    assert torch.equal(ndl, -delta) and torch.equal(nl2, -lse * torch.tensor(1.4426950408889634, dtype=torch.float32, device=lse.device))
    # the backward with and without its own delta pass
    dk0, dqkv0 = torch.empty_like(k), torch.zeros_like(qkv_d)
    dqw0 = K.attn_bwd_qnorm(q, k, v, ctx, plain, lse, B, S, Hq, Hkv, D, dk0, dqkv0[:, (Hq + Hkv) * D :], qkv_d, qw_d, cos_d, sin_d, pos_d, rstd, dqkv0, causal=True)
    fused2 = K.dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D)  # the scratch was emptied by the backward above: fill it again
    dk1, dqkv1 = torch.empty_like(k), torch.zeros_like(qkv_d)
    dqw1 = K.attn_bwd_qnorm(q, k, v, ctx, fused2[0], lse, B, S, Hq, Hkv, D, dk1, dqkv1[:, (Hq + Hkv) * D :], qkv_d, qw_d, cos_d, sin_d, pos_d, rstd, dqkv1, causal=True,
                            delta=fused2[1])
    assert torch.equal(fused2[1], delta), "bit-reproducible"
    nq = Hq * D
    assert rel_l2(dk1, dk0) < 1e-3 and rel_l2(dqkv1[:, :nq], dqkv0[:, :nq]) < 1e-3 and rel_l2(dqw1, dqw0) < 1e-3
    assert torch.equal(dqkv1[:, (Hq + Hkv) * D :], dqkv0[:, (Hq + Hkv) * D :]), "dV does not depend on delta"
    # shapes the form does not take fall back (None): few rows
    assert K.dgrad_attn_delta(dy[: 2 * S], w, ctx[: 2 * S], lse[:2].contiguous(), 2, S, Hq, D) is None


def test_attention_strided_views(K):
    """k/v read in place from the fused QKV projection buffer (row pitch = (Hq+2Hkv)*D)."""
    B, S, Hq, Hkv, D = 1, 100, 4, 2, 128
    g = torch.Generator().manual_seed(4)
    qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, generator=g).to(BF16)
    dq_ = dev(qkv)
    q, k, v = dq_[:, : Hq * D], dq_[:, Hq * D : (Hq + Hkv) * D], dq_[:, (Hq + Hkv) * D :]
    o, _ = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D)
    o_ref, _ = _attn_ref(qkv[:, : Hq * D], qkv[:, Hq * D : (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D :], B, S, Hq, Hkv, D, None, True)
    assert rel_l2(o, o_ref) < 4e-3


def test_attention_fully_masked_rows_follow_reference(K):
    """Left padding: queries whose visible keys are all padding attend uniformly over ALL keys upstream
    (finite fill value, qwen3_attention.py:139-142)."""
    B, S, Hq, Hkv, D = 1, 200, 2, 1, 128
    g = torch.Generator().manual_seed(8)
    q = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    k = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    v = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    km = torch.ones(B, S, dtype=torch.uint8)
    km[0, :70] = 0
    o_ref, _ = _attn_ref(q, k, v, B, S, Hq, Hkv, D, km, True)
    o, _ = K.attn_fwd(dev(q), dev(k), dev(v), B, S, Hq, Hkv, D, key_mask=dev(km), causal=True)
    assert rel_l2(o, o_ref) < 4e-3


# ---- the lean-softmax forward (csrc/attention.hip, the default) against fp64 and against the first-generation kernel
def _forward_kernel(K, bit):
    """Context manager: ablation bit 11 (2048) keeps the first-generation forward kernel (0 = the default lean kernel)."""
    import contextlib

    @contextlib.contextmanager
    def cm():
        keep = K._ATTN_ABLATE
        K._ATTN_ABLATE = keep | (bit << 8)
        try:
            yield
        finally:
            K._ATTN_ABLATE = keep

    return cm()


def _old_forward(K):
    return _forward_kernel(K, 2048)


FWD2_CASES = [
    # B, S, Hq, Hkv, causal, mask ("none" | "right" = padded tail | "left" = padded head: fully masked rows | "holes")
    (3, 709, 4, 2, True, "none"),
    (2, 1024, 8, 2, True, "none"),     # four query heads per kv head = two pairs
    (2, 300, 2, 1, False, "none"),
    (1, 129, 2, 1, True, "right"),
    (2, 709, 4, 2, True, "left"),
    (2, 450, 2, 2, True, "none"),      # one query head per kv head: not this kernel's shape, the first-generation kernel runs
    (5, 64, 2, 1, True, "none"),
    (1, 1, 2, 1, True, "none"),
    (2, 63, 4, 2, False, "holes"),
    (40, 200, 16, 8, True, "right"),   # 160 head pairs x 2 blocks = 320 items on 256 workgroups: the tile stream crosses item boundaries
]


@pytest.mark.parametrize("kernel", ["lean"])
@pytest.mark.parametrize("B,S,Hq,Hkv,causal,mask", FWD2_CASES)
def test_attention_forward_second_generation_matches_fp64(K, B, S, Hq, Hkv, causal, mask, kernel):
    """Output and log-sum-exp against fp64 softmax with the reference's finite mask fill (qwen3_attention.py:130-142), over shapes that walk
    one and several blocks per item, both tile-count parities, padded tails, padded heads (rows whose visible keys are all masked) and holes."""
    D = 128
    g = torch.Generator().manual_seed(S * 7 + Hq)
    q = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    k = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    v = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    km = None
    if mask != "none":
        km = torch.ones(B, S, dtype=torch.uint8)
        if mask == "right":
            km[0, S - S // 3 :] = 0
        elif mask == "left":
            km[0, :70] = 0
            km[1, :1] = 0
        else:
            km[:, ::5] = 0
            km[1, 0] = 1
    big = B * Hq * S * S > 3e7  # keep the fp64 reference on the device when the score tensor is large
    rd = (lambda t: t.cuda()) if big else (lambda t: t)
    o_ref, p_ref = _attn_ref(rd(q), rd(k), rd(v), B, S, Hq, Hkv, D, None if km is None else rd(km), causal)
    with _forward_kernel(K, 0):
        o, lse = K.attn_fwd(dev(q), dev(k), dev(v), B, S, Hq, Hkv, D, key_mask=None if km is None else dev(km), causal=causal)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    assert rel_l2(o, o_ref) < 4e-3, rel_l2(o, o_ref)
    # lse = log sum exp of the masked scaled scores (the backward recomputes P from it)
    q4 = rd(q).double().view(B, S, Hq, D).transpose(1, 2)
    k4 = rd(k).double().view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, dim=1)
    sc = (q4 @ k4.mT) * D ** -0.5
    blocked = torch.zeros(B, 1, S, S, dtype=torch.bool, device=sc.device)
    if causal:
        blocked = blocked | torch.triu(torch.ones(S, S, dtype=torch.bool, device=sc.device), 1)
    if km is not None:
        blocked = blocked | ~rd(km).bool()[:, None, None, :]
    fill = -2.0e38 * 0.6931471805599453  # the kernel's finite fill in natural-log units
    lse_ref = torch.logsumexp(sc.masked_fill(blocked, fill), dim=-1)
    live = ~blocked.expand(B, Hq, S, S).all(dim=-1)  # rows with at least one visible key (the others sit at the fill value)
    # (the scale is folded into the bf16 query rows: one more rounding of an operand, ~1e-3 on a score)
    assert float((lse.double().cpu() - lse_ref.cpu())[live.cpu()].abs().max()) < 6e-3
    with _old_forward(K):
        o1, lse1 = K.attn_fwd(dev(q), dev(k), dev(v), B, S, Hq, Hkv, D, key_mask=None if km is None else dev(km), causal=causal)
    assert rel_l2(o, o1) < 5e-3 and float((lse - lse1)[live.to(lse.device)].abs().max()) < 6e-3


@pytest.mark.parametrize("kernel", ["lean"])
def test_attention_forward_lazy_rescale_branch_is_exercised(K, kernel):
    """The forward rescales O only when a row maximum outgrows its running reference by 2^8 -- a rare, data-dependent branch
    that bounded random data never takes.  Keys aligned with chosen queries make the maximum jump by ~30 log2 units at chosen tiles (first
    tile, an interior tile, the block's last tile, for either head of the pair); every output row is checked against fp64."""
    B, S, Hq, Hkv, D = 2, 709, 4, 2, 128
    g = torch.Generator().manual_seed(77)
    q = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    k = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    v = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    q4, k4 = q.view(B, S, Hq, D), k.view(B, S, Hkv, D)
    for b, row, key, hq in ((0, 600, 400, 0), (0, 100, 70, 1), (1, 700, 690, 2), (1, 330, 5, 3), (0, 640, 639, 0), (1, 64, 63, 1)):
        k4[b, key, hq // 2] = (2.0 * q4[b, row, hq].float()).to(BF16)
    o_ref, p_ref = _attn_ref(q, k, v, B, S, Hq, Hkv, D, None, True)
    assert float(p_ref.view(B, Hq, S, S)[0, 0, 600, 400]) > 0.99  # the spike really dominates its row
    with _forward_kernel(K, 0):
        o, lse = K.attn_fwd(dev(q), dev(k), dev(v), B, S, Hq, Hkv, D, causal=True)
    assert rel_l2(o, o_ref) < 4e-3
    err = (o.float().cpu() - o_ref.float()).view(B, S, Hq * D).norm(dim=-1) / o_ref.float().view(B, S, Hq * D).norm(dim=-1)
    assert float(err.max()) < 2e-2, (float(err.max()), int(err.argmax()))
    # and the backward, which recomputes P from this forward's lse, still matches
    do = torch.randn(B * S, Hq * D, generator=g).to(BF16)
    qr, kr, vr = (t.double().requires_grad_(True) for t in (q, k, v))
    o_r, _ = _attn_ref(qr, kr, vr, B, S, Hq, Hkv, D, None, True)
    o_r.backward(do.double())
    dq, dk, dv = (torch.zeros_like(dev(t)) for t in (q, k, v))
    K.attn_bwd(dev(q), dev(k), dev(v), o, dev(do), lse, B, S, Hq, Hkv, D, dq, dk, dv, causal=True)
    for name, got, ref in (("dq", dq, qr.grad), ("dk", dk, kr.grad), ("dv", dv, vr.grad)):
        assert rel_l2(got, ref) < 8e-3, name


# ----------------------------------------------------------------------------------------------- CE / embedding / gathers
def test_cross_entropy_golden(K, golden):
    t = golden("per_op")
    lg, tg = t["ce.logits"], t["ce.targets"]
    loss_rows, _ = K.cross_entropy(dev(lg), dev(tg), want_grad=False)
    out3 = K.ce_finalize(loss_rows, dev(tg)).cpu()
    ref32 = torch.nn.functional.cross_entropy(lg.float(), tg, ignore_index=-100)
    assert abs(float(out3[0]) - float(ref32)) < 1e-5 * abs(float(ref32))
    assert float(out3[0].to(BF16)) == float(t["ce.loss"])  # the reference's bf16 scalar
    assert int(out3[1]) == 10
    scale = torch.tensor([1.0 / 10], dtype=F32)
    _, dl = K.cross_entropy(dev(lg.clone()), dev(tg), want_grad=True, grad_scale=dev(scale), inplace=False)
    assert rel_l2(dl, t["ce.glogits"]) < 4e-3
    assert torch.count_nonzero(dl[[2, 7]]) == 0


def test_cross_entropy_large_vocab(K):
    g = torch.Generator().manual_seed(31)
    rows, V = 64, 151_936
    lg = (torch.randn(rows, V, generator=g) * 2).to(BF16)
    tg = torch.randint(0, V, (rows,), generator=g)
    tg[5] = -100
    loss_rows, _ = K.cross_entropy(dev(lg), dev(tg), want_grad=False)
    ref = torch.nn.functional.cross_entropy(lg.float(), tg, ignore_index=-100, reduction="none")
    assert rel_l2(loss_rows, ref) < 1e-5


@pytest.mark.parametrize("V", [1000, 8200, 32768, 32776, 151_936, 155_648, 155_656, 5003])
@pytest.mark.parametrize("inplace", [True, False])
def test_cross_entropy_forms_agree_with_fp32(K, V, inplace):
    """Every dispatch of mi355_cross_entropy: the two-read kernel (V <= 8192, V > 155 648, V % 8 != 0) and the row-in-registers
    kernel in both chunk counts (4 and 19 per thread); ignore_index rows, an out-of-range target, the target in the row's first
    and last chunk, in place and out of place.  Loss rows to 1e-5 of the fp32 value, gradient to bf16 rounding (ref engine.py:45,60)."""
    g = torch.Generator().manual_seed(V)
    rows = 9
    lg = (torch.randn(rows, V, generator=g) * 3).to(BF16)
    tg = torch.randint(0, V, (rows,), generator=g)
    tg[0], tg[1], tg[2], tg[3] = 0, V - 1, -100, V + 5  # first / last element, ignored, out of range
    scale = torch.tensor([0.125], dtype=F32)
    pitch = (V + 7) // 8 * 8  # rows are 16-byte aligned (the entry point refuses anything else): an odd vocabulary is a view into padded rows
    x = torch.zeros(rows, pitch, dtype=BF16, device="cuda")[:, :V]
    x.copy_(lg)
    loss_rows, dl = K.cross_entropy(x, dev(tg), want_grad=True, grad_scale=dev(scale), inplace=inplace)
    assert (dl.data_ptr() == x.data_ptr()) == inplace
    ok = [r for r in range(rows) if r not in (2, 3)]
    lref = lg.float().requires_grad_(True)
    ref = torch.nn.functional.cross_entropy(lref[ok], tg[ok], reduction="none")
    ref.sum().backward()
    assert float((loss_rows.cpu()[ok] - ref.detach()).abs().max()) < 1e-5 * float(ref.detach().abs().max())
    assert float(loss_rows[2]) == 0.0 and torch.isnan(loss_rows[3])
    assert torch.count_nonzero(dl[[2, 3]]) == 0
    assert rel_l2(dl.cpu()[ok], lref.grad[ok] * 0.125) < 4e-3
    if not inplace:
        assert torch.equal(x.cpu(), lg)  # the logits are untouched


def test_embedding_and_scatter(K):
    g = torch.Generator().manual_seed(41)
    table = torch.randn(1000, 1024, generator=g).to(BF16)
    ids = torch.randint(0, 1000, (3, 70), generator=g)
    out = K.embedding_fwd(dev(ids), dev(table))
    assert torch.equal(out.cpu(), table[ids.reshape(-1)])  # bit-exact gather
    dout = torch.randn(210, 1024, generator=g).to(BF16)
    acc = torch.zeros(1000, 1024, dtype=F32, device="cuda")
    K.embedding_bwd(dev(ids), dev(dout), acc)
    ref = torch.zeros(1000, 1024).index_add_(0, ids.reshape(-1), dout.float())
    assert rel_l2(acc, ref) < 1e-6


def test_early_fusion_copy_bit_exact(K, golden):
    t = golden("index_ops")
    vis, txt = t["fusion.vis"].to(BF16), t["fusion.txt"].to(BF16)
    B, nv, w = vis.shape
    nt = txt.shape[1]
    fused = torch.empty(B, nv + nt, w, dtype=BF16, device="cuda")
    for b in range(B):
        K.copy2d(dev(vis[b]), fused[b, :nv])
        K.copy2d(dev(txt[b]), fused[b, nv:])
    assert torch.equal(fused.cpu(), t["fusion.cat"].to(BF16))
    # batched form: one strided copy per source
    big = torch.empty(B * (nv + nt), w, dtype=BF16, device="cuda")
    K.copy2d(dev(vis).view(B, nv * w), big.view(B, (nv + nt) * w)[:, : nv * w])
    K.copy2d(dev(txt).view(B, nt * w), big.view(B, (nv + nt) * w)[:, nv * w :])
    assert torch.equal(big.view(B, nv + nt, w).cpu(), t["fusion.cat"].to(BF16))


def test_patchify_bit_exact(K, golden):
    from oracle import index_ops

    t = golden("index_ops")
    for name, (hw, p) in {"p32_4": (32, 4), "p224_16": (224, 16)}.items():
        img = torch.arange(2 * 3 * hw * hw, dtype=F32).view(2, 3, hw, hw)
        rows = K.patchify(dev(img), p, out_dtype=F32).cpu()
        idx = torch.from_numpy(index_ops.patch_gather_index(3, hw, hw, p))
        assert np.array_equal(idx.numpy(), t[f"patch_gather.{name}"].numpy().astype(np.int64))
        for b in range(2):
            npatch = idx.shape[0]
            assert torch.equal(rows[b * npatch : (b + 1) * npatch], img[b].reshape(-1)[idx])
    img = torch.randn(2, 3, 32, 32)
    rows = K.patchify(dev(img), 8, out_dtype=BF16).cpu()
    idx = torch.from_numpy(index_ops.patch_gather_index(3, 32, 32, 8))
    assert torch.equal(rows[:16], img[0].reshape(-1)[idx].to(BF16))


def test_cast_clip_helpers(K):
    g = torch.Generator().manual_seed(51)
    x = torch.randn(100_003, generator=g)
    xb = K.cast(dev(x), BF16)
    assert torch.equal(xb.cpu(), x.to(BF16))
    assert torch.equal(K.cast(xb, F32).cpu(), x.to(BF16).float())
    acc = torch.zeros(1, dtype=F32, device="cuda")
    K.sumsq_into(xb, acc)
    K.sumsq_into(dev(x), acc)
    ref = float(x.to(BF16).float().pow(2).sum() + x.pow(2).sum())
    assert abs(float(acc) - ref) < 1e-4 * ref
    y = dev(x.clone())
    ss = torch.tensor([float(x.pow(2).sum())], dtype=F32, device="cuda")
    K.clip_scale_(y, ss, 1.0)
    coef = 1.0 / (math.sqrt(float(x.pow(2).sum())) + 1e-6)
    assert rel_l2(y, x * coef) < 1e-6


def test_gemm_swiglu_backward_epilogue_equals_two_kernels_bit_for_bit():
    """d(gate-up) from ONE dgrad GEMM with the SwiGLU-backward epilogue == dgrad GEMM -> bf16 d(act) -> mi355_swiglu_bwd, at the step's
    shape (ragged M) and at a small one."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    torch.manual_seed(21)
    for M, F, Nout in ((709 * 3 + 5, 3072, 1024), (300, 256, 64)):
        dy = torch.randn(M, Nout, device="cuda").to(torch.bfloat16)
        w = (torch.randn(Nout, F, device="cuda") / Nout**0.5).to(torch.bfloat16)
        gu = torch.randn(M, 2 * F, device="cuda").to(torch.bfloat16)
        want = K.swiglu_bwd(gu, K.gemm(L.GEMM_NN, dy, w, allow_split_k=False), F)
        got = K.gemm_dgrad_swiglu_bwd(dy, w, gu)
        assert torch.equal(got, want)


def test_gemm_swiglu_forward_epilogue_equals_two_kernels_bit_for_bit():
    """(gate-up output, activation) from ONE projection GEMM with the SwiGLU epilogue (weight rows fetched as [32 lin1 | 32 lin_gate] groups)
    == projection GEMM -> mi355_swiglu_fwd, for every tile configuration that serves NT, ragged M, F not a multiple of the tile."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    torch.manual_seed(22)
    for M, F, Kd in ((709 * 2 + 3, 3072, 1024), (300, 3584, 256), (130, 96, 64)):
        x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
        w = (torch.randn(2 * F, Kd, device="cuda") / Kd**0.5).to(torch.bfloat16)
        gu_ref = K.gemm(L.GEMM_NT, x, w, allow_split_k=False)
        a_ref = K.swiglu_fwd(gu_ref, F)
        for tile in (0, 1, 2, 3):
            gu, a = K.gemm_gateup_swiglu(x, w, tile=tile)
            assert torch.equal(gu, gu_ref), (M, F, tile)
            assert torch.equal(a, a_ref), (M, F, tile)


@pytest.mark.parametrize("M,N,K_", [(256, 256, 128), (1000, 512, 128), (709 * 5 + 3, 1024, 1024), (2600, 4096, 256), (4099, 264, 192), (300, 256, 64), (513, 776, 2048), (4200, 512, 256)])
def test_persistent_nt_kernel_equals_the_per_tile_kernel_bit_for_bit(M, N, K_, monkeypatch):
    """Tile hint 7 (one workgroup per CU walking its tiles, K-tile stream across tile boundaries, packed-bf16 write-out) against hint 2 for every epilogue it
    carries -- plain, residual, SwiGLU forward, SwiGLU backward -- on ragged M, N that is no multiple of the tile, a single tile, more tiles than CUs;
    K % 64 != 0 and K < 128 fall back to the per-tile kernel inside the library.  Then the same through the automatic choice (threshold lowered to one tile)."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    g = torch.Generator().manual_seed(M + 3 * N + 7 * K_)
    rnd = lambda *s: dev((0.3 * torch.randn(*s, generator=g)).to(BF16))
    x, w, res = rnd(M, K_), rnd(N, K_), rnd(M, N)
    ref = K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False)
    ref_res = K.gemm(L.GEMM_NT, x, w, residual=res, tile=2, allow_split_k=False)
    got = torch.full_like(ref, 7.0)  # every element must be written
    K.gemm(L.GEMM_NT, x, w, out=got, tile=7)
    assert torch.equal(got, ref)
    assert torch.equal(K.gemm(L.GEMM_NT, x, w, residual=res, tile=7), ref_res)
    F = (N // 64) * 32  # fused gate-up weight [2F, K], F % 32 == 0
    wgu = rnd(2 * F, K_)
    gu2, a2 = K.gemm_gateup_swiglu(x, wgu, tile=2)
    gu7, a7 = K.gemm_gateup_swiglu(x, wgu, tile=7)
    assert torch.equal(gu7, gu2) and torch.equal(a7, a2)
    dy, w2, gu = rnd(M, K_), rnd(K_, N), rnd(M, 2 * N)
    assert torch.equal(K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=7), K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=2))
    monkeypatch.setenv("MI355_GEMM_PERSIST_MIN_TILES", "1")
    assert torch.equal(K.gemm(L.GEMM_NT, x, w, allow_split_k=False), ref)
    assert torch.equal(K.gemm(L.GEMM_NT, x, w, residual=res, allow_split_k=False), ref_res)
    gu0, a0 = K.gemm_gateup_swiglu(x, wgu)
    assert torch.equal(gu0, gu2) and torch.equal(a0, a2)


@pytest.mark.parametrize("M,N,K_", [(709 * 5 + 3, 1024, 2048), (2600, 4096, 512), (4099, 264, 384), (513, 776, 2048), (300, 256, 320), (2049, 1288, 640), (70000, 1536, 128)])
def test_weight_stationary_walk_equals_the_per_tile_kernel_bit_for_bit(M, N, K_, monkeypatch):
    """Round 6: the persistent NT kernel's weight-stationary XCD walk (tile hint 7 with ablation bit 3; the library's own choice by shape, MI355_GEMM_WALK) changes only
    WHICH workgroup computes a tile and when: every epilogue gives the per-tile kernel's bits on ragged M, N that is no multiple of the tile or of a column group (empty
    tiles are walked as zero work), fewer pairs than row streams, more tiles than CUs.  Then the same through the automatic choice with the walk forced on / off."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    W = 7 + (8 << 8)
    g = torch.Generator().manual_seed(M + 3 * N + 7 * K_)
    rnd = lambda *s: dev((0.3 * torch.randn(*s, generator=g)).to(BF16))
    x, w, res = rnd(M, K_), rnd(N, K_), rnd(M, N)
    ref = K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False)
    got = torch.full_like(ref, 7.0)  # every element must be written
    K.gemm(L.GEMM_NT, x, w, out=got, tile=W)
    assert torch.equal(got, ref)
    assert torch.equal(K.gemm(L.GEMM_NT, x, w, residual=res, tile=W), K.gemm(L.GEMM_NT, x, w, residual=res, tile=2, allow_split_k=False))
    F = (N // 64) * 32
    wgu = rnd(2 * F, K_)
    gu2, a2 = K.gemm_gateup_swiglu(x, wgu, tile=2)
    guw, aw = K.gemm_gateup_swiglu(x, wgu, tile=W)
    assert torch.equal(guw, gu2) and torch.equal(aw, a2)
    dy, w2, gu = rnd(M, K_), rnd(K_, N), rnd(M, 2 * N)
    assert torch.equal(K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=W), K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=2))
    monkeypatch.setenv("MI355_GEMM_PERSIST_MIN_TILES", "1")
    for mode in ("1", "0", "2"):
        monkeypatch.setenv("MI355_GEMM_WALK", mode)
        assert torch.equal(K.gemm(L.GEMM_NT, x, w, allow_split_k=False), ref), mode


@pytest.mark.parametrize("M,N,K_", [(709 * 5 + 3, 1024, 1024), (4099, 264, 384), (513, 776, 2048), (300, 256, 320), (5000, 3072, 1024)])
def test_ping_pong_nt_kernel_is_exact_on_exact_products_and_within_fp32_noise_otherwise(M, N, K_):
    """Round 6, tile hint 8 (gemm_nt_pp_kernel; opt-in, not the library's choice): the two wave groups run a write-out apart on one stream, so a tile's K-tiles are summed in a
    ROTATED order.  On small-integer operands every product and partial sum is exact: the same bits as tile 2 for every epilogue.  On random operands the fp32 sums differ
    in their last bits: a bf16 output may move by one unit in the last place where a sum sits on a rounding boundary (< 0.1 % of the elements), never more."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    g = torch.Generator().manual_seed(M + N + K_)
    ints = lambda *s: dev(torch.randint(-2, 3, s, generator=g).float().to(BF16))
    rnd = lambda *s: dev((0.3 * torch.randn(*s, generator=g)).to(BF16))

    def close(a, b, slack=1.01):
        af, bf = a.float(), b.float()
        d = (af - bf).abs()
        unit = torch.maximum(bf.abs() * 2.0**-7, 1e-4 * bf.pow(2).mean().sqrt())
        return float((d / unit).max()) <= slack and float((d > 0).float().mean()) < 1e-3

    for mk, same in ((ints, torch.equal), (rnd, close)):
        x, w, res = mk(M, K_), mk(N, K_), mk(M, N)
        got = torch.full((M, N), 7.0, dtype=BF16, device="cuda")
        K.gemm(L.GEMM_NT, x, w, out=got, tile=8)
        assert same(got, K.gemm(L.GEMM_NT, x, w, tile=2, allow_split_k=False))
        assert same(K.gemm(L.GEMM_NT, x, w, residual=res, tile=8), K.gemm(L.GEMM_NT, x, w, residual=res, tile=2, allow_split_k=False))
        F = (N // 64) * 32
        wgu = mk(2 * F, K_)
        gu8, a8 = K.gemm_gateup_swiglu(x, wgu, tile=8)
        gu2, a2 = K.gemm_gateup_swiglu(x, wgu, tile=2)
        assert same(gu8, gu2) and (torch.equal(a8, a2) if mk is ints else close(a8, a2, slack=8.0))  # (a moved gate-up bit moves the activation's last bits too)
        dy, w2, gu = mk(M, K_), mk(K_, N), rnd(M, 2 * N)
        b8, b2 = K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=8), K.gemm_dgrad_swiglu_bwd(dy, w2, gu, tile=2)
        assert torch.equal(b8, b2) if mk is ints else close(b8, b2, slack=8.0)


@pytest.mark.parametrize("tanh", [False, True])
def test_gemm_gelu_epilogues_equal_two_kernels_bit_for_bit(tanh):
    """Linear -> GELU (pre-activation + activation from one launch) and the GELU backward in the next Linear's dgrad epilogue == the
    GEMM + elementwise-kernel forms, erf and tanh."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    K._FUSE_GELU = True  # the fused forms are an option (the separate kernels are faster on the ViT step and stay the default)
    torch.manual_seed(23)
    for M, Kd, F in ((197 * 5 + 3, 768, 3072), (260, 64, 264)):
        x = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
        w1 = (torch.randn(F, Kd, device="cuda") / Kd**0.5).to(torch.bfloat16)
        b1 = torch.randn(F, device="cuda")
        y_ref = K.gemm(L.GEMM_NT, x, w1, bias=b1, allow_split_k=False)
        f_ref = K.gelu_fwd(y_ref, tanh=tanh)
        y1, f = K.gemm_gelu_dual(x, w1, bias=b1, tanh=tanh)
        assert torch.equal(y1, y_ref) and torch.equal(f, f_ref)
        dy = torch.randn(M, Kd, device="cuda").to(torch.bfloat16)
        w2 = (torch.randn(Kd, F, device="cuda") / F**0.5).to(torch.bfloat16)
        want = K.gelu_bwd(y_ref, K.gemm(L.GEMM_NN, dy, w2, allow_split_k=False), tanh=tanh)
        assert torch.equal(K.gemm_dgrad_gelu_bwd(dy, w2, y_ref, tanh=tanh), want)
    K._FUSE_GELU = False


def test_embedding_backward_sorted_is_deterministic_and_matches_fp32(K):
    """``mi355_embedding_bwd_sorted`` (the gradient of ``emb_dict``, reference qwen3_model.py:69 through autograd): heavy id repetition (the
    case fp32 atomics make irreproducible), ids outside the table, a row-strided gradient buffer.  Against an fp32 index_add; bit-identical
    between runs; rows no id names keep their bits; the non-accumulating form overwrites exactly the named rows."""
    V, W, T = 1000, 1024, 6000
    g = torch.Generator().manual_seed(3)
    ids = torch.randint(0, 40, (T,), generator=g)  # ~150 tokens per id
    ids[::7] = torch.randint(0, V, (len(ids[::7]),), generator=g)
    ids[5], ids[77] = -100, V + 3  # skipped
    dout = torch.randn(T, W + 64, generator=g).to(BF16)[:, :W]  # row pitch W + 64
    table0 = torch.randn(V, W, generator=g).to(BF16)
    ok = (ids >= 0) & (ids < V)
    ref = table0.float().index_add(0, ids[ok], 0.5 * dout[ok].float())
    outs = []
    for _ in range(2):
        t = dev(table0.clone())
        K.embedding_bwd_sorted(dev(ids), dev(dout), t, True, 0.5)
        outs.append(t)
    assert torch.equal(outs[0], outs[1])
    named = torch.zeros(V, dtype=torch.bool)
    named[ids[ok]] = True
    assert torch.equal(outs[0].cpu()[~named], table0[~named])
    assert rel_l2(outs[0][dev(named)], ref[named]) < 3e-3
    t = dev(table0.clone())
    K.embedding_bwd_sorted(dev(ids), dev(dout), t, False, 1.0)
    ref2 = torch.zeros(V, W).index_add(0, ids[ok], dout[ok].float())
    assert torch.equal(t.cpu()[~named], table0[~named]) and rel_l2(t[dev(named)], ref2[named]) < 3e-3
    # every permutation of the same (id, row) pairs that keeps equal ids in order gives the same bits: the sum runs in token order
    perm = torch.argsort(ids, stable=True)
    t2 = dev(table0.clone())
    K.embedding_bwd_sorted(dev(ids[perm]), dev(dout[perm].contiguous()), t2, False, 1.0)
    assert torch.equal(t2, t)


@pytest.mark.parametrize("T", [4099, 64, 33, 31])
def test_embedding_backward_sorted_long_runs_span_blocks(K, T):
    """Padding / placeholder tokens: one id thousands of times.  The sum is a two-level tree over blocks of 32 sorted positions (pass 1 per block,
    pass 2 joins the parts of a run that crosses blocks): runs that start and end on and off block boundaries, a run that fills whole blocks, the
    last (short) block, ids outside the table inside a long run's neighbourhood.  Against fp64, bit-identical between launches."""
    V, W = 300, 512
    g = torch.Generator().manual_seed(T)
    ids = torch.randint(0, V, (T,), generator=g)
    ids[torch.rand(T, generator=g) < 0.7] = 17  # ~70 % of the positions: one run over many blocks
    if T > 200:
        ids[100:164] = 250  # 64 positions: two full blocks' worth, wherever the sort puts them
        ids[200:233] = 251  # 33 positions
        ids[7] = -5
    dout = torch.randn(T, W, generator=g).to(BF16)
    ok = (ids >= 0) & (ids < V)
    ref = torch.zeros(V, W, dtype=torch.float64).index_add(0, ids[ok], dout[ok].double())
    outs = []
    for _ in range(2):
        t = dev(torch.full((V, W), 7.0).to(BF16))
        K.embedding_bwd_sorted(dev(ids), dev(dout), t, False, 1.0)
        outs.append(t)
    assert torch.equal(outs[0], outs[1])
    named = torch.zeros(V, dtype=torch.bool)
    named[ids[ok]] = True
    got = outs[0].cpu()
    assert torch.equal(got[~named], torch.full((V, W), 7.0).to(BF16)[~named])
    err = (got[named].double() - ref[named]).abs().max() / ref[named].abs().max()
    assert err < 6e-3, float(err)  # one bf16 rounding of sums of up to thousands of rows
