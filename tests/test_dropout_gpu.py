"""Dropout on the HIP path (reference sites: vit_model.py:146, vit_attention.py:79, vit_transformer_block.py:117,124, vit_engine.py:51).

No implementation shares torch's random stream, so parity is split as ``oracle/dropout.py`` explains: the masks are Bernoulli(1 - p)
(keep rate inside 3 sigma) and reproducible from (seed, offset); GIVEN the masks -- rebuilt on the CPU by the oracle's Philox4x32-10
restatement, itself pinned by the published known-answer vectors in tests/test_dropout_cpu.py -- every kernel equals the reference
arithmetic: bit-exact for the element-wise sites, stated tolerances for attention and for the whole ViT training step."""

import math

import pytest
import torch

from conftest import sub_dict
from oracle import dropout as OD
from oracle import models as OM
from oracle.gen_golden import TINY_VIT

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


@pytest.mark.parametrize("n", [4096, 1003, 3])
def test_elementwise_dropout_is_bit_exact_given_the_mask(n):
    from llm_quest_amd import kernels as K

    torch.manual_seed(0)
    p, seed, off = 0.1, 0x1234_5678_9ABC, 7
    mul = OD.elementwise_multiplier((n,), p, seed, off)
    x = torch.randn(n)
    y = K.dropout(x.cuda(), p, seed, off).cpu()
    assert torch.equal(y, x * mul)
    res = torch.randn(n)
    assert torch.equal(K.dropout(x.cuda(), p, seed, off, residual=res.cuda()).cpu(), res + x * mul)
    xb = x.to(BF16)
    assert torch.equal(K.dropout(xb.cuda(), p, seed, off).cpu(), (xb.float() * mul).to(BF16))
    # the backward form: fp32 gradient in, bf16 out, same mask
    assert torch.equal(K.dropout(x.cuda(), p, seed, off, out_dtype=BF16).cpu(), (x * mul).to(BF16))
    # another offset is another mask; p = 0 keeps everything
    if n > 100:
        assert not torch.equal(K.dropout(x.cuda(), p, seed, off + 1).cpu(), y)
    assert torch.equal(K.dropout(x.cuda(), 0.0, seed, off).cpu(), x)


def test_keep_rate_is_bernoulli():
    from llm_quest_amd import kernels as K

    n = 1 << 22
    x = torch.ones(n, device="cuda")
    for p in (0.1, 0.5, 0.03):
        kept = float((K.dropout(x, p, 99, 3) != 0).float().mean())
        assert abs(kept - (1 - p)) < 3 * math.sqrt(p * (1 - p) / n), (p, kept)
    B, H, S = 2, 3, 197
    q = torch.zeros(B * S, H * 64, dtype=BF16, device="cuda")  # zero scores: uniform weights 1/S, so O = mean of the kept V rows / (1-p)
    v = torch.ones(B * S, H * 64, dtype=BF16, device="cuda")
    o, _ = K.attn_dropout_fwd(q, q, v, B, S, H, H, 64, 0.1, 5, 11)
    kept = float(o.float().mean()) * 0.9  # each output = (#kept / S) / 0.9
    assert abs(kept - 0.9) < 3 * math.sqrt(0.1 * 0.9 / (B * H * S * S)) + 2e-3  # + bf16 rounding of the outputs


@pytest.mark.parametrize("causal", [False, True])
@pytest.mark.parametrize("D,S,H,Hkv", [(64, 197, 3, 3), (32, 50, 4, 2), (128, 130, 2, 1)])
def test_attention_dropout_matches_reference_given_the_mask(D, S, H, Hkv, causal):
    """softmax -> dropout -> @ V (vit_attention.py:74-81) in fp32 torch ops with the oracle's mask, forward and all three gradients."""
    from llm_quest_amd import kernels as K

    torch.manual_seed(1)
    B, p, seed, off = 2, 0.1, 4242, 9
    q = torch.randn(B * S, H * D).to(BF16)
    k = torch.randn(B * S, Hkv * D).to(BF16)
    v = torch.randn(B * S, Hkv * D).to(BF16)
    g = torch.randn(B * S, H * D).to(BF16)
    o, lse = K.attn_dropout_fwd(q.cuda(), k.cuda(), v.cuda(), B, S, H, Hkv, D, p, seed, off, causal=causal)
    dq, dk, dv = torch.empty_like(q).cuda(), torch.empty_like(k).cuda(), torch.empty_like(v).cuda()
    K.attn_dropout_bwd(q.cuda(), k.cuda(), v.cuda(), o, g.cuda(), lse, B, S, H, Hkv, D, dq, dk, dv, p, seed, off, causal=causal)
    mul = OD.attention_multiplier(B, H, S, p, seed, off)
    qr, kr, vr = (t.float().requires_grad_(True) for t in (q, k, v))
    rep = H // Hkv
    q4 = qr.view(B, S, H, D).transpose(1, 2)
    k4 = kr.view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(rep, dim=1)
    v4 = vr.view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(rep, dim=1)
    sc = (q4 @ k4.mT) * D**-0.5
    if causal:
        sc = sc.masked_fill(torch.triu(torch.ones(S, S, dtype=torch.bool), 1), float("-inf"))
    ref = ((torch.softmax(sc, -1) * mul) @ v4).transpose(1, 2).reshape(B * S, H * D)
    ref.backward(g.float())
    assert rel_l2(o, ref) < 8e-3
    assert rel_l2(lse, torch.logsumexp(sc, -1)) < 1e-5
    assert rel_l2(dq, qr.grad) < 1.2e-2 and rel_l2(dk, kr.grad) < 1.2e-2 and rel_l2(dv, vr.grad) < 1.2e-2
    # p = 0 on these kernels is plain attention
    o0, _ = K.attn_dropout_fwd(q.cuda(), k.cuda(), v.cuda(), B, S, H, Hkv, D, 0.0, seed, off, causal=causal)
    ref0 = (torch.softmax(sc, -1) @ v4).transpose(1, 2).reshape(B * S, H * D)
    assert rel_l2(o0, ref0) < 8e-3


def _tiny_vit(golden, drop):
    from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel

    t = golden("vit_tiny")
    m = ViTModel(dict(TINY_VIT, drop_rate=drop))
    m.load_state_dict(sub_dict(t, "sd."))
    return t, m.cuda().train()


def test_vit_training_step_with_dropout_matches_the_oracle_given_the_masks(golden):
    """ViTModel(drop_rate=0.1).train(): logits, loss and every parameter gradient against the oracle's ViT (the reference's forward,
    oracle/models.py::vit_forward) evaluated in fp32 with the SAME masks, at the tolerances of the drop_rate = 0 test."""
    from llm_quest_amd import rng
    from llm_quest_amd.engine import _cross_entropy

    p, seed = 0.1, 2024
    t, m = _tiny_vit(golden, p)
    img, y = t["in.image"], t["in.labels"]
    B, S, d, H, Ln = img.shape[0], 17, TINY_VIT["emb_dim"], TINY_VIT["n_heads"], TINY_VIT["n_layers"]
    rng.manual(seed, 0)
    try:
        logits = m(img.cuda())
        loss = _cross_entropy(logits, y.cuda())
        loss.backward()
        rng.manual(seed, 0)
        again = m(img.cuda())
        other = m(img.cuda())  # offsets moved on: different masks
    finally:
        rng.follow_torch()
    assert torch.equal(again, logits) and not torch.equal(other, logits)
    # draw order of the forward: embedding, then per block attention weights, projection output, FFN output
    drop = {"embed": OD.elementwise_multiplier((B, S, d), p, seed, 0), "blocks": []}
    for i in range(Ln):
        drop["blocks"].append({"att": OD.attention_multiplier(B, H, S, p, seed, 1 + 3 * i), "proj": OD.elementwise_multiplier((B, S, d), p, seed, 2 + 3 * i),
                               "ffn": OD.elementwise_multiplier((B, S, d), p, seed, 3 + 3 * i)})
    sd = {k: v.clone().requires_grad_(True) for k, v in sub_dict(t, "sd.").items()}
    ref_logits = OM.vit_forward(sd, TINY_VIT, img, drop=drop)
    ref_loss = torch.nn.functional.cross_entropy(ref_logits, y)
    ref_loss.backward()
    assert rel_l2(logits, ref_logits) < 2e-2
    assert abs(float(loss.detach()) - float(ref_loss.detach())) / float(ref_loss.detach()) < 5e-3
    for name, prm in m.named_parameters():
        ref = sd[name].grad.double()
        err = float((prm.grad.double().cpu() - ref).norm())
        assert err <= 3e-2 * float(ref.norm()) + 2e-4, f"{name}: |err| {err:.3e}, |ref| {float(ref.norm()):.3e}"
    # the masks really acted: the no-dropout forward differs
    assert rel_l2(ref_logits, OM.vit_forward(sub_dict(t, "sd."), TINY_VIT, img)) > 1e-2
    # eval mode ignores drop_rate; train mode without autograd still drops (forward-only path)
    with torch.no_grad():
        assert rel_l2(m.eval()(img.cuda()), t["out.logits"]) < 2e-2
        for prm in m.parameters():
            prm.requires_grad = False
        rng.manual(seed, 0)
        try:
            frozen = m.train()(img.cuda())
        finally:
            rng.follow_torch()
        assert rel_l2(frozen, ref_logits) < 2e-2


def test_stock_vit_base_config_trains_with_its_dropout():
    """BASELINE config 2 as configured upstream: VIT_BASE_CONFIG has drop_rate = 0.1 (config.py:170)."""
    from llm_quest_amd.config import VIT_BASE_CONFIG
    from llm_quest_amd.engine import _cross_entropy
    from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel

    assert VIT_BASE_CONFIG["drop_rate"] == 0.1
    torch.manual_seed(0)
    m = ViTModel(dict(VIT_BASE_CONFIG)).cuda().train()
    img = torch.randn(8, 3, 224, 224, device="cuda")
    y = torch.randint(0, VIT_BASE_CONFIG["num_classes"], (8,), device="cuda")
    loss = _cross_entropy(m(img), y)
    loss.backward()
    assert math.isfinite(float(loss)) and abs(float(loss) - math.log(VIT_BASE_CONFIG["num_classes"])) < 1.5
    assert all(p.grad is not None and bool(torch.isfinite(p.grad).all()) for p in m.parameters())


def test_adapter_dropout_given_the_mask():
    from llm_quest_amd import rng
    from llm_quest_amd.multimodal.vision_transformer.vit_engine import ViTAdapter

    torch.manual_seed(3)
    p, seed = 0.2, 77
    ad = ViTAdapter(64, 128, adapter_type="ffn", dropout=p, dtype=BF16).cuda().train()
    x = torch.randn(2, 9, 64).to(BF16).cuda().requires_grad_(True)
    g = torch.randn(2, 9, 128).to(BF16).cuda()
    rng.manual(seed, 4)
    try:
        y = ad(x)
    finally:
        rng.follow_torch()
    y.backward(g)
    mul = OD.elementwise_multiplier((2, 9, 256), p, seed, 4)
    w0, w3 = ad.adapter[0].weight.detach().float().cpu(), ad.adapter[3].weight.detach().float().cpu()
    xr = x.detach().float().cpu().requires_grad_(True)
    h = torch.nn.functional.gelu(torch.nn.functional.linear(xr, w0).to(BF16).float()).to(BF16).float() * mul
    yr = torch.nn.functional.linear(h.to(BF16).float(), w3)
    yr.backward(g.float().cpu())
    assert rel_l2(y, yr) < 1e-2 and rel_l2(x.grad, xr.grad) < 2e-2
    assert float((ad.eval()(x.detach()) - y).abs().max()) > 0  # eval: no dropout


def test_dropout_stream_follows_the_device_generator_and_leaves_the_cpu_generator_alone():
    """On the GPU the (seed, offset) pairs come from the current device's default generator, as a device ``nn.Dropout`` takes them: the SAME
    seed set again restarts the masks, ``torch.cuda.get_rng_state`` / ``set_rng_state`` resume them, torch's CPU generator is never consumed
    (llm_quest_amd/rng.py; reference sites vit_model.py:146, vit_attention.py:79)."""
    from llm_quest_amd import kernels as K
    from llm_quest_amd import rng

    torch.cuda.init()
    rng.follow_torch()
    torch.manual_seed(1234)
    cpu_before = torch.get_rng_state().clone()
    a = [rng.draw() for _ in range(4)]
    assert torch.equal(cpu_before, torch.get_rng_state())
    assert all(s == 1234 for s, _ in a) and len({o for _, o in a}) == 4
    torch.manual_seed(1234)
    assert [rng.draw() for _ in range(4)] == a
    st = torch.cuda.get_rng_state()
    c = [rng.draw() for _ in range(2)]
    torch.cuda.set_rng_state(st)
    assert [rng.draw() for _ in range(2)] == c
    x = torch.ones(1 << 16, device="cuda", dtype=BF16)
    torch.manual_seed(99)
    y1 = K.dropout(x, 0.5, *rng.draw())
    y2 = K.dropout(x, 0.5, *rng.draw())
    torch.manual_seed(99)
    assert torch.equal(K.dropout(x, 0.5, *rng.draw()), y1) and not torch.equal(y1, y2)
