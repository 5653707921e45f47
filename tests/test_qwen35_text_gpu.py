"""BASELINE config 5 text stack on HIP kernels vs the CPU oracle (oracle/qwen3_5_text.py, itself
pinned to fixtures generated from the reference): kernel level first, then layers and the tiny hybrid model.
Bit-exact: index gathers.  <= 1 bf16 ulp: elementwise ops whose rounding points are reproduced.  Stated relative-L2 tolerances
elsewhere (bf16 operands, fp32 accumulation in a different order than ATen's)."""

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import sub_dict
from oracle import qwen3_5_text as OT

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def ulp_diff(a, b):
    """max distance in bf16 ulps between two bf16 tensors (monotone integer view)."""
    def key(t):
        i = t.cpu().contiguous().view(torch.int16).to(torch.int32)
        return torch.where(i < 0, -(i & 0x7FFF), i)
    return int((key(a) - key(b)).abs().max())


def tok_major(x):  # (b, h, s, d) -> [b*s, h*d]
    b, h, s, d = x.shape
    return x.permute(0, 2, 1, 3).reshape(b * s, h * d).contiguous()


def head_major(x2d, b, s, h):  # [b*s, h*d] -> (b, h, s, d)
    return x2d.reshape(b, s, h, -1).permute(0, 2, 1, 3)


# ------------------------------------------------------------------------------------------------- small ops
def test_zero_centered_rmsnorm_matches_fixture(golden):
    from llm_quest_amd import kernels as K
    from llm_quest_amd import kernels_q35 as Q

    t = golden("qwen35_text_tiny")
    x, scale = t["zc.x"], t["zc.scale"]
    w = Q.zc_weight(scale.cuda())
    assert torch.equal(w.cpu(), (1.0 + scale))  # bf16(1 + scale), the reference's own rounding
    y, _ = K.rmsnorm_fwd(x.reshape(-1, 64).cuda().contiguous(), w)
    assert ulp_diff(y.view(x.shape), t["zc.out"]) <= 1
    # real width
    torch.manual_seed(0)
    x = torch.randn(37, 1024).to(BF16)
    sc = (0.1 * torch.randn(1024)).to(BF16)
    y, _ = K.rmsnorm_fwd(x.cuda(), Q.zc_weight(sc.cuda()))
    assert ulp_diff(y, OT.zc_rmsnorm(x, sc)) <= 1


def test_mrope_table_is_an_exact_gather(golden):
    from llm_quest_amd import kernels_q35 as Q

    t = golden("qwen35_text_tiny")
    cos, sin, pid = t["mrope.cos"], t["mrope.sin"], t["mrope.pid"]
    ct, st = Q.mrope_table(cos.cuda(), sin.cuda(), pid.cuda(), [3, 3, 2])
    rc, rs = OT.mrope_coeffs(cos, sin, pid, [3, 3, 2])
    assert torch.equal(ct.cpu().view(2, 10, -1), rc) and torch.equal(st.cpu().view(2, 10, -1), rs)
    # real section sizes, rotation dim 64
    torch.manual_seed(1)
    cos, sin = torch.randn(512, 64), torch.randn(512, 64)
    cos[:, 32:], sin[:, 32:] = cos[:, :32], sin[:, :32]
    pid = torch.randint(0, 512, (3, 3, 41))
    ct, st = Q.mrope_table(cos.cuda(), sin.cuda(), pid.cuda(), [11, 11, 10])
    rc, rs = OT.mrope_coeffs(cos, sin, pid, [11, 11, 10])
    assert torch.equal(ct.cpu().view(3, 41, -1), rc) and torch.equal(st.cpu().view(3, 41, -1), rs)


@pytest.mark.parametrize("H,D,R,stride_mul", [(8, 256, 64, 2), (2, 256, 64, 1), (2, 32, 16, 2), (3, 128, 64, 1)])
def test_headnorm_rope_forward_backward(H, D, R, stride_mul):
    from llm_quest_amd import kernels_q35 as Q

    torch.manual_seed(2)
    b, s = 2, 19
    tokens = b * s
    hs = D * stride_mul
    proj = torch.randn(tokens, H * hs + 24).to(BF16)  # heads strided inside a wider projection
    scale = (0.2 * torch.randn(D)).to(BF16)
    cos_t, sin_t = torch.randn(tokens, R), torch.randn(tokens, R)
    cos_t[:, R // 2 :], sin_t[:, R // 2 :] = cos_t[:, : R // 2], sin_t[:, : R // 2]
    pos = torch.arange(tokens, dtype=torch.int32)
    x = torch.stack([proj[:, h * hs : h * hs + D] for h in range(H)], 1).view(b, s, H, D).transpose(1, 2)  # (b,H,s,D)

    def ref(xx, sc, dt):
        n = OT.zc_rmsnorm(xx, sc) if dt == BF16 else (xx * torch.rsqrt(xx.pow(2).mean(-1, keepdim=True) + 1e-6) * (1.0 + sc))
        c, sn = cos_t.view(b, 1, s, R).to(dt), sin_t.view(b, 1, s, R).to(dt)
        return OT.rope_partial(n, c, sn)

    want = ref(x, scale, BF16)
    dproj = proj.cuda()
    w = Q.zc_weight(scale.cuda())
    out, rstd = Q.headnorm_rope_fwd(dproj[:, : H * hs], H, D, hs, w, cos_t.cuda(), sin_t.cuda(), pos.cuda())
    got = head_major(out, b, s, H)
    assert ulp_diff(got, want) <= 1
    # backward vs fp32 autograd of the same function (bf16-rounded coefficients)
    xf = x.float().requires_grad_(True)
    sf = (1.0 + scale).float() - 1.0  # so that (1 + sf) == the bf16-rounded factor the kernel uses
    sf.requires_grad_(True)
    cb, sb = cos_t.to(BF16).float().view(b, 1, s, R), sin_t.to(BF16).float().view(b, 1, s, R)
    n = xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + 1e-6) * (1.0 + sf)
    y = OT.rope_partial(n, cb, sb)
    g = torch.randn(b, H, s, D).to(BF16)
    y.backward(g.float())
    dsrc = torch.zeros_like(dproj)
    dw = Q.headnorm_rope_bwd(dproj[:, : H * hs], H, D, hs, w, cos_t.cuda(), sin_t.cuda(), pos.cuda(), rstd, tok_major(g).cuda(), dsrc[:, : H * hs], hs)
    gx = torch.stack([dsrc.cpu()[:, h * hs : h * hs + D] for h in range(H)], 1).view(b, s, H, D).transpose(1, 2)
    assert rel_l2(gx, xf.grad) < 6e-3  # bf16 rounding of the written gradient
    assert rel_l2(dw, sf.grad) < 1e-4
    if stride_mul == 2:  # the gate halves between the heads are not touched
        for h in range(H):
            assert not dsrc.cpu()[:, h * hs + D : (h + 1) * hs].any()


def test_sigmoid_gate():
    from llm_quest_amd import kernels_q35 as Q

    torch.manual_seed(3)
    tokens, H, D = 45, 8, 256
    proj = torch.randn(tokens, H * 2 * D + 16).to(BF16)
    ctx = torch.randn(tokens, H * D).to(BF16)
    gate = proj[:, : H * 2 * D].view(tokens, H, 2 * D)[..., D:].reshape(tokens, H * D)
    want = ctx * torch.sigmoid(gate)
    dproj = proj.cuda()
    out = Q.sigmoid_gate_fwd(ctx.cuda(), dproj[:, D:], H, D, 2 * D)
    assert ulp_diff(out, want) <= 1
    g = torch.randn(tokens, H * D).to(BF16)
    cf, gf = ctx.float().requires_grad_(True), gate.float().requires_grad_(True)
    (cf * torch.sigmoid(gf)).backward(g.float())
    dbuf = torch.zeros_like(dproj)
    dctx = Q.sigmoid_gate_bwd(ctx.cuda(), dproj[:, D:], H, D, 2 * D, g.cuda(), dbuf[:, D:], 2 * D)
    assert rel_l2(dctx, cf.grad) < 6e-3
    dgate = dbuf.cpu()[:, : H * 2 * D].view(tokens, H, 2 * D)[..., D:].reshape(tokens, H * D)
    assert rel_l2(dgate, gf.grad) < 6e-3
    assert not dbuf.cpu()[:, : H * 2 * D].view(tokens, H, 2 * D)[..., :D].any()


def test_gdn_gates():
    from llm_quest_amd import kernels_q35 as Q

    torch.manual_seed(4)
    tokens, Hv = 333, 16
    proj = (2.0 * torch.randn(tokens, 64)).to(BF16)
    log_A = torch.log(torch.rand(Hv) * 16)
    dtb = (1.0 + 0.1 * torch.randn(Hv)).to(BF16)
    b_lin, a_lin = proj[:, 24 : 24 + Hv], proj[:, 40 : 40 + Hv]
    beta_ref = torch.sigmoid(b_lin)  # bf16
    alpha_ref = OT.alpha_factor(log_A, a_lin, dtb)  # fp32 (bf16 add, bf16 softplus)
    dproj = proj.cuda()
    beta, alpha = Q.gdn_gates_fwd(dproj[:, 24 : 24 + Hv], dproj[:, 40 : 40 + Hv], log_A.cuda(), dtb.cuda())
    assert ulp_diff(beta.to(BF16), beta_ref) <= 1 and torch.equal(beta.cpu(), beta.cpu().to(BF16).float())
    assert alpha.dtype == F32 and rel_l2(alpha, alpha_ref) < 1e-6
    # backward vs fp32 autograd
    bl, al = b_lin.float().requires_grad_(True), a_lin.float().requires_grad_(True)
    la, db = log_A.clone().requires_grad_(True), dtb.float().requires_grad_(True)
    gb, ga = torch.randn(tokens, Hv), torch.randn(tokens, Hv)
    (torch.sigmoid(bl) * gb).sum().backward()
    (OT.alpha_factor(la, al, db) * ga).sum().backward()
    dbuf = torch.zeros_like(dproj)
    dla, ddb = Q.gdn_gates_bwd(dproj[:, 24 : 24 + Hv], dproj[:, 40 : 40 + Hv], log_A.cuda(), dtb.cuda(), gb.cuda(), ga.cuda(), dbuf[:, 24 : 24 + Hv], dbuf[:, 40 : 40 + Hv])
    assert rel_l2(dbuf[:, 24 : 24 + Hv], bl.grad) < 6e-3
    assert rel_l2(dbuf[:, 40 : 40 + Hv], al.grad) < 2e-2  # softplus input / output are bf16-rounded in the forward
    assert rel_l2(dla, la.grad) < 2e-2 and rel_l2(ddb, db.grad) < 2e-2


@pytest.mark.parametrize("B,S,C", [(2, 37, 6144), (3, 5, 128), (1, 2, 64)])
def test_causal_conv_silu(B, S, C):
    from llm_quest_amd import kernels_q35 as Q

    torch.manual_seed(5)
    proj = torch.randn(B * S, C + 40).to(BF16)
    w = (0.5 * torch.randn(C, 1, 4)).to(BF16)
    x = proj[:, :C].reshape(B, S, C)
    want32 = OT.causal_depthwise_conv_silu(x.float(), w.float())
    want = OT.causal_depthwise_conv_silu(x, w)  # bf16 ATen (its accumulation order is its own)
    dproj = proj.cuda()
    y = Q.causal_conv_silu_fwd(dproj[:, :C], w.cuda(), B, S)
    assert rel_l2(y, want32.reshape(B * S, C)) < 5e-3
    assert ulp_diff(y, want.reshape(B * S, C)) <= 4
    # backward vs fp32 autograd
    xf, wf = x.float().requires_grad_(True), w.float().requires_grad_(True)
    g = torch.randn(B, S, C).to(BF16)
    OT.causal_depthwise_conv_silu(xf, wf).backward(g.float())
    dbuf = torch.zeros_like(dproj)
    dw = Q.causal_conv_silu_bwd(dproj[:, :C], w.cuda(), g.reshape(B * S, C).cuda(), dbuf[:, :C], B, S)
    assert rel_l2(dbuf[:, :C], xf.grad.reshape(B * S, C)) < 1e-2
    assert rel_l2(dw.view(C, 1, 4), wf.grad) < 1e-2
    assert not dbuf.cpu()[:, C:].any()


def test_l2norm():
    from llm_quest_amd import kernels_q35 as Q

    torch.manual_seed(6)
    tokens, H, D = 77, 16, 128
    buf = torch.randn(tokens, H * D + 64).to(BF16)
    buf[3, :D] = 0  # a zero vector: the clamped branch
    x = buf[:, : H * D].view(tokens, H, D)
    want = OT.l2_norm(x)
    dbuf_in = buf.cuda()
    y = Q.l2norm_fwd(dbuf_in[:, : H * D], H, D)
    assert ulp_diff(y.view(tokens, H, D), want) <= 1
    xf = x.float().requires_grad_(True)
    g = torch.randn(tokens, H, D).to(BF16)
    OT.l2_norm(xf).backward(g.float())
    dx = torch.zeros_like(dbuf_in)
    Q.l2norm_bwd(dbuf_in[:, : H * D], g.reshape(tokens, H * D).cuda(), dx[:, : H * D], H, D)
    assert rel_l2(dx[:, : H * D], xf.grad.reshape(tokens, H * D)) < 6e-3


def test_gated_rmsnorm():
    from llm_quest_amd import kernels_q35 as Q
    from oracle import ops as OO

    torch.manual_seed(7)
    tokens, H, D = 91, 16, 128
    o = torch.randn(tokens, H * D).to(BF16)
    proj = torch.randn(tokens, H * D + 32).to(BF16)
    w = 1.0 + 0.1 * torch.randn(D)
    gate = proj[:, 32:]

    def ref(of, wf, gf):
        xn = of.view(tokens, H, D)
        xn = xn * torch.rsqrt(xn.pow(2).mean(-1, keepdim=True) + 1e-6) * wf
        return F.silu(gf) * xn.reshape(tokens, H * D)

    want = ref(o.float(), w, gate.float()).to(BF16)
    dproj = proj.cuda()
    out, rstd = Q.gated_rmsnorm_fwd(o.cuda(), w.cuda(), dproj[:, 32:], H, D)
    assert ulp_diff(out, want) <= 1
    of, wf, gf = o.float().requires_grad_(True), w.clone().requires_grad_(True), gate.float().requires_grad_(True)
    g = torch.randn(tokens, H * D).to(BF16)
    ref(of, wf, gf).backward(g.float())
    dbuf = torch.zeros_like(dproj)
    d_o, dw = Q.gated_rmsnorm_bwd(o.cuda(), w.cuda(), dproj[:, 32:], rstd, g.cuda(), dbuf[:, 32:], H, D)
    assert rel_l2(d_o, of.grad) < 6e-3 and rel_l2(dbuf[:, 32:], gf.grad) < 6e-3 and rel_l2(dw, wf.grad) < 1e-4


# ------------------------------------------------------------------------------------------------- gated delta rule
def _gdr_run(q, k, v, beta, alpha, go):
    """q,k: (b,hqk,s,dk) bf16; v: (b,hv,s,dv) bf16; beta/alpha (b,hv,s) fp32; go (b,hv,s,dv) fp32.  HIP forward + backward."""
    from llm_quest_amd import kernels_q35 as Q

    b, hqk, s, dk = q.shape
    hv, dv = v.shape[1], v.shape[3]
    qd, kd = tok_major(q).cuda(), tok_major(k).cuda()
    vbuf = torch.zeros(b * s, hv * dv + 16, dtype=BF16)
    vbuf[:, : hv * dv] = tok_major(v)
    vbuf = vbuf.cuda()
    be = beta.permute(0, 2, 1).reshape(b * s, hv).contiguous().cuda()
    al = alpha.permute(0, 2, 1).reshape(b * s, hv).contiguous().cuda()
    o, ck, fin = Q.gated_delta_rule_fwd(qd, kd, vbuf[:, : hv * dv], be, al, b, s, hqk, hv, dk, dv, keep=True, want_state=True)
    dvbuf = torch.zeros_like(vbuf)
    dq, dk_, dbe, dal = Q.gated_delta_rule_bwd(qd, kd, vbuf[:, : hv * dv], be, al, ck, tok_major(go.to(BF16)).cuda(), dvbuf[:, : hv * dv], b, s, hqk, hv, dk, dv)
    back = lambda t, h: head_major(t.cpu(), b, s, h)
    return (back(o, hv), fin.cpu(), back(dq, hqk), back(dk_, hqk), back(dvbuf[:, : hv * dv], hv),
            dbe.cpu().view(b, s, hv).permute(0, 2, 1), dal.cpu().view(b, s, hv).permute(0, 2, 1))


def test_gated_delta_rule_matches_reference_fixture(golden):
    t = golden("qwen35_text_tiny")
    q, k, v, beta, alpha, go = (t["gdr." + n] for n in ("q", "k", "v", "beta", "alpha", "gout"))
    o, fin, dq, dk, dv, dbeta, dalpha = _gdr_run(q, k, v, beta, alpha, go.to(BF16).float())
    assert ulp_diff(o, t["gdr.out"]) <= 1
    assert rel_l2(fin, t["gdr.state"]) < 1e-5
    # the fixture's upstream gradient is fp32; ours enters as bf16 -> bf16-level agreement
    for got, name in ((dq, "q"), (dk, "k"), (dv, "v"), (dbeta, "beta"), (dalpha, "alpha")):
        assert rel_l2(got, t["gdr.grad." + name]) < 8e-3, name


def test_gated_delta_rule_real_head_dims_three_chunks():
    torch.manual_seed(8)
    b, hqk, hv, s, dk, dv = 2, 2, 4, 150, 128, 128  # 3 checkpoint chunks, value heads share q/k heads in pairs
    q = OT.l2_norm(torch.randn(b, hqk, s, dk)).to(BF16)
    k = OT.l2_norm(torch.randn(b, hqk, s, dk)).to(BF16)
    v = torch.randn(b, hv, s, dv).to(BF16)
    beta = torch.rand(b, hv, s).to(BF16).float()
    alpha = 0.3 + 0.7 * torch.rand(b, hv, s)
    go = torch.randn(b, hv, s, dv).to(BF16).float()
    o, fin, dq, dk_, dv_, dbeta, dalpha = _gdr_run(q, k, v, beta, alpha, go)
    qf, kf, vf = (x.float().requires_grad_(True) for x in (q, k, v))
    bf_, af = beta.clone().requires_grad_(True), alpha.clone().requires_grad_(True)
    ro, rstate = OT.gated_delta_rule(qf.repeat_interleave(2, 1), kf.repeat_interleave(2, 1), vf, bf_, af)
    assert rel_l2(o, ro) < 4e-3  # bf16 rounding of the output; sums of 128 products in another order than ATen
    assert rel_l2(fin, rstate) < 1e-5
    (ro * go).sum().backward()
    assert rel_l2(dq, qf.grad) < 6e-3 and rel_l2(dk_, kf.grad) < 6e-3 and rel_l2(dv_, vf.grad) < 6e-3
    assert rel_l2(dbeta, bf_.grad) < 1e-4 and rel_l2(dalpha, af.grad) < 1e-4


# ------------------------------------------------------------------------------------------------- attention (SDPA semantics)
@pytest.mark.parametrize("B,S,Hq,Hkv,D,masked", [(2, 300, 8, 2, 256, False), (2, 173, 8, 2, 256, True), (2, 40, 2, 1, 32, True), (1, 130, 4, 2, 128, False), (1, 65, 2, 2, 64, False)])
def test_attention_generic_forward_backward(B, S, Hq, Hkv, D, masked):
    from llm_quest_amd import kernels_q35 as Q

    torch.manual_seed(9)
    q = torch.randn(B, Hq, S, D).to(BF16)
    k = torch.randn(B, Hkv, S, D).to(BF16)
    v = torch.randn(B, Hkv, S, D).to(BF16)
    am = None
    allow = torch.tril(torch.ones(S, S, dtype=torch.bool))
    if masked:
        am = torch.ones(B, S, dtype=torch.bool)
        am[0, S - 7 :] = False
        am[B - 1, S // 2 : S // 2 + 3] = False  # upstream ORs ~attn_mask into the allow mask: these keys become visible to all
        allow = allow.view(1, 1, S, S) | ~am.view(B, 1, 1, S)
    qf, kf, vf = (x.float().requires_grad_(True) for x in (q, k, v))
    ref = F.scaled_dot_product_attention(qf, kf, vf, attn_mask=allow, enable_gqa=True)
    go = torch.randn(B, Hq, S, D).to(BF16)
    ref.backward(go.float())
    qd, kd, vd = tok_major(q).cuda(), tok_major(k).cuda(), tok_major(v).cuda()
    km = None if am is None else am.to(torch.uint8).cuda()
    o, lse = Q.attn_generic_fwd(qd, kd, vd, B, S, Hq, Hkv, D, key_mask=km)
    assert rel_l2(head_major(o, B, S, Hq), ref) < 6e-3
    dq, dk, dv = torch.empty_like(qd), torch.empty_like(kd), torch.empty_like(vd)
    Q.attn_generic_bwd(qd, kd, vd, o, tok_major(go).cuda(), lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=km)
    assert rel_l2(head_major(dq, B, S, Hq), qf.grad) < 1e-2
    assert rel_l2(head_major(dk, B, S, Hkv), kf.grad) < 1e-2
    assert rel_l2(head_major(dv, B, S, Hkv), vf.grad) < 1e-2
    # determinism
    o2, _ = Q.attn_generic_fwd(qd, kd, vd, B, S, Hq, Hkv, D, key_mask=km)
    assert torch.equal(o, o2)
