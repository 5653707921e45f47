"""Drop-in boundary entry points that run on their own: ``RoPE.apply`` / ``apply_mrope`` / ``VisionRoPE.apply`` on
device tensors (``mi355_rope_apply``) and the ``forward`` of every Qwen3.5 vision sub-module, against fixtures the reference produced
(``tests/golden/rope_extra.safetensors``, ``per_op``, ``qwen35_text_tiny``, ``qwen35_vision_tiny``)."""

import pytest
import torch

from conftest import sub_dict
from oracle.gen_golden import TINY_Q35_VISION

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel_l2(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_rope_apply_on_device_is_bit_exact(golden):
    """bf16: coefficients, both products and the sum round to bf16 exactly where the reference's torch ops do -> identical bits;
    fp32: plain fp32 multiplies and one add, no contraction -> identical bits."""
    from llm_quest_amd.common.rope import RoPE, VisionRoPE

    t = golden("per_op")
    x = t["rope.x"].cuda()
    assert torch.equal(RoPE.apply(x, t["rope.cos"].cuda(), t["rope.sin"].cuda()).cpu(), t["rope.y"])
    assert torch.equal(RoPE.apply(x, t["rope.cos"], t["rope.sin"], t["rope.pid"].cuda()).cpu(), t["rope.y_pid"])  # host tables are moved
    e = golden("rope_extra")
    cos, sin, pid = e["part.cos"].cuda(), e["part.sin"].cuda(), e["part.pid"].cuda()
    for tag in ("bf16", "fp32"):
        xx = e[f"part.{tag}.x"].cuda().requires_grad_(True)
        y = RoPE.apply(xx, cos, sin, pid)  # 32 of 64 features rotate, the rest pass through
        assert y.dtype == xx.dtype and torch.equal(y.cpu(), e[f"part.{tag}.y"]), tag
        y.backward(e[f"part.{tag}.g"].cuda())
        assert torch.equal(xx.grad.cpu(), e[f"part.{tag}.gx"]), tag
    # token-major heads handed over as a transposed view (no copy: the kernel takes the strides)
    xt = e["tm.x"].cuda()
    yt = RoPE.apply(xt.transpose(1, 2), cos, sin)
    assert yt.stride() == xt.transpose(1, 2).stride() and torch.equal(yt.contiguous().cpu(), e["tm.y"])
    # 2-D axial tables over two frames
    assert torch.equal(VisionRoPE.apply(e["vis.x"].cuda(), e["vis.cos"].cuda(), e["vis.sin"].cuda()).cpu(), e["vis.y"])
    # an odd half width takes the scalar path of the kernel
    c6, s6 = RoPE.compute_angles(10_000, 12, 8, rotation_factor=0.5)  # 6 features rotate: half = 3
    x6 = torch.randn(1, 2, 8, 12).to(BF16)
    assert torch.equal(RoPE.apply(x6.cuda(), c6.cuda(), s6.cuda()).cpu(), RoPE.apply(x6, c6, s6))


def test_apply_mrope_on_device_is_bit_exact(golden):
    from llm_quest_amd.common.rope import RoPE

    t = golden("qwen35_text_tiny")
    out = RoPE.apply_mrope(t["mrope.q"].cuda(), t["mrope.cos"].cuda(), t["mrope.sin"].cuda(), t["mrope.pid"].cuda(), [3, 3, 2])
    assert torch.equal(out.cpu(), t["mrope.out"])
    e = golden("rope_extra")
    q = e["mrope.q"].cuda().requires_grad_(True)
    y = RoPE.apply_mrope(q, e["mrope.cos"].cuda(), e["mrope.sin"].cuda(), e["mrope.pid"].cuda(), [3, 3, 2])
    assert torch.equal(y.cpu(), e["mrope.y"])
    y.backward(e["mrope.g"].cuda())
    assert torch.equal(q.grad.cpu(), e["mrope.gq"])


def _tower(golden):
    from llm_quest_amd.qwen.qwen3_5.qwen3_5_vision_model import Qwen3_5VisionModel

    t = golden("qwen35_vision_tiny")
    m = Qwen3_5VisionModel(dict(TINY_Q35_VISION))
    m.load_state_dict(sub_dict(t, "vis.sd."))
    return t, m.cuda().train()


def test_vision_submodule_forwards_compose_to_the_reference_tower(golden):
    """PatchEmbedding3D.forward -> + pos_embed -> Qwen3_5VisionTransformerBlock.forward(x, cos, sin) x L -> ViTMergeAdapter.forward, each
    called through its own ``forward`` as a user of the reference would, reproduces the reference tower's output and every parameter
    gradient at the tolerances of the whole-model test, and agrees with this package's one-node tower."""
    t, m = _tower(golden)
    pixels = t["vis.in"].cuda()
    x = m.patch_embed(pixels)
    nsp = m.n_spatial_patches
    frames = x.shape[1] // nsp
    assert x.shape == (pixels.shape[0], frames * nsp, m.emb_dim) and x.dtype == F32
    x = x + m.pos_embed(torch.arange(nsp, device="cuda")).unsqueeze(0).repeat(1, frames, 1)
    cos, sin = m.cos.repeat(frames, 1), m.sin.repeat(frames, 1)
    for blk in m.blocks:
        x = blk(x, cos, sin)
    out = m.merge_adapter(x)
    assert out.shape == t["vis.out"].shape
    assert rel_l2(out, t["vis.out"]) < 1.5e-2
    out.backward(t["vis.gout"].cuda())
    ref = sub_dict(t, "vis.grad.")
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        err = float((p.grad.double().cpu() - ref[name].double()).norm())
        assert err <= 4e-2 * float(ref[name].double().norm()) + 1e-3, f"{name}: |err| {err:.3e} |ref| {float(ref[name].norm()):.3e}"
    with torch.no_grad():
        assert rel_l2(out, m(pixels)) < 2e-3  # the tower fuses the positional add into the patch GEMM's epilogue


def test_vision_attention_and_ffn_forward_alone(golden):
    """Qwen3_5VisionAttention.forward(x, cos, sin) and Qwen3_5VisionFFN.forward(x) against the same arithmetic in fp32 torch ops on the
    module's own weights (the reference's forward bodies, qwen3_5_vision_model.py:124-125, 153-198), forward and input gradient."""
    from llm_quest_amd.common.rope import VisionRoPE

    t, m = _tower(golden)
    blk = m.blocks[0]
    torch.manual_seed(5)
    B, S, d = 2, 2 * m.n_spatial_patches, m.emb_dim
    x = torch.randn(B, S, d, device="cuda").requires_grad_(True)
    g = torch.randn(B, S, d, device="cuda")
    cos, sin = m.cos.repeat(2, 1), m.sin.repeat(2, 1)
    # --- attention
    y = blk.att(x, cos, sin)
    y.backward(g)
    gx, x.grad = x.grad.clone(), None
    att, H, Dh = blk.att, blk.att.num_heads, blk.att.head_dim
    xr = x.detach().clone().requires_grad_(True)
    q, k, v = torch.nn.functional.linear(xr, att.qkv.weight, att.qkv.bias).chunk(3, dim=-1)
    q, k, v = (u.view(B, S, H, Dh).transpose(1, 2) for u in (q, k, v))
    rot = lambda u: cos[:S] * u + sin[:S] * torch.cat((-u[..., Dh // 2 :], u[..., : Dh // 2]), dim=-1)
    ctx = torch.nn.functional.scaled_dot_product_attention(rot(q), rot(k), v)
    yr = torch.nn.functional.linear(ctx.transpose(1, 2).reshape(B, S, d), att.proj.weight, att.proj.bias)
    yr.backward(g)
    assert y.dtype == F32 and rel_l2(y, yr) < 1.5e-2 and rel_l2(gx, xr.grad) < 2e-2
    assert att.qkv.weight.grad is not None and att.proj.bias.grad is not None
    # --- FFN (tanh GELU)
    y2 = blk.ffn(x)
    y2.backward(g)
    ffn = blk.ffn
    xr2 = x.detach().clone().requires_grad_(True)
    yr2 = torch.nn.functional.linear(torch.nn.functional.gelu(torch.nn.functional.linear(xr2, ffn.lin1.weight, ffn.lin1.bias), approximate="tanh"), ffn.lin2.weight, ffn.lin2.bias)
    yr2.backward(g)
    assert rel_l2(y2, yr2) < 1.5e-2 and rel_l2(x.grad, xr2.grad) < 2e-2
    # bf16 input gives a bf16 result, eval / no_grad works
    with torch.no_grad():
        assert blk.ffn(x.detach().to(BF16)).dtype == BF16


# ------------------------------------------------------------------------------------------- ViT sub-modules, stand-alone (row b)
def _vit(golden):
    from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel
    from oracle.gen_golden import TINY_VIT

    t = golden("vit_tiny")
    m = ViTModel(dict(TINY_VIT))
    m.load_state_dict(sub_dict(t, "sd."))
    return t, m.cuda().train(), TINY_VIT


def test_vit_submodule_forwards_compose_to_the_reference_gradients(golden):
    """PatchEmbedding2D.forward -> + pos_embedding -> ViTTransformerBlock.forward x L -> LayerNorm.forward -> classifier, each module called
    through its OWN forward (reference vit_model.py:68-89, vit_transformer_block.py:22-31,106-127) as ordinary autograd modules: the
    composition reproduces the reference model's logits, loss and every parameter gradient of the ``vit_tiny`` fixture."""
    t, m, cfg = _vit(golden)
    img, y = t["in.image"].cuda(), t["in.labels"].cuda()
    h = m.patch_embedding(img)
    assert h.requires_grad and h.shape == (img.shape[0], m.patch_embedding.num_patches + 1, cfg["emb_dim"])
    h = h + m.pos_embedding
    for blk in m.transformer_blocks:
        h = blk(h)
        assert h.requires_grad and h.dtype == F32
    hn = m.final_ln(h)
    assert rel_l2(hn, t["out.hidden"]) < 1e-2
    logits = torch.nn.functional.linear(hn[:, 0], m.classifier.weight, m.classifier.bias)  # 10 x 64: the test's own glue, not the product path
    assert rel_l2(logits, t["out.logits"]) < 2e-2
    loss = torch.nn.functional.cross_entropy(logits.float(), y)
    assert abs(float(loss) - float(t["out.loss"])) / float(t["out.loss"]) < 5e-3
    loss.backward()
    ref = sub_dict(t, "grad.")
    for name, p in m.named_parameters():
        assert p.grad is not None, name
        err = float((p.grad.double().cpu() - ref[name].double()).norm())
        assert err <= 3e-2 * float(ref[name].double().norm()) + 2e-4, f"{name}: |err| {err:.3e}, |ref| {float(ref[name].norm()):.3e}"


def test_vit_attention_ffn_layernorm_gelu_train_stand_alone(golden):
    """ViTMultiHeadAttention / FFN / LayerNorm / GELU on their own (vit_attention.py:44-91, vit_transformer_block.py:22-67): forward, input
    gradient and every parameter gradient against the oracle's fp32 restatement on the same weights (3e-2 = bf16 MFMA operands)."""
    from oracle import models, ops

    t, m, cfg = _vit(golden)
    blk = m.transformer_blocks[1]
    sd = {k: v.detach().float().cpu().requires_grad_(True) for k, v in blk.state_dict().items()}
    g = torch.Generator().manual_seed(5)
    x = torch.randn(3, 17, cfg["emb_dim"], generator=g)
    gy = torch.randn(3, 17, cfg["emb_dim"], generator=g)

    def check(mod, pfx, out, xg, ref_out, xr):
        assert rel_l2(out, ref_out) < 2e-2, pfx
        assert rel_l2(xg, xr.grad) < 3e-2, pfx + " dx"
        # w_keys.bias has a mathematically ZERO gradient (softmax ignores a constant key offset): absolute slack on the scale of the module's
        # other gradients (bf16 rounding of dS), relative tolerance elsewhere
        scale = max(float(sd[pfx + n].grad.double().norm()) for n, _ in mod.named_parameters())
        for name, p in mod.named_parameters():
            r = sd[pfx + name].grad
            err = float((p.grad.double().cpu() - r.double()).norm())
            assert err <= 3e-2 * float(r.double().norm()) + 2e-3 * scale, f"{pfx}{name}: {err:.3e} vs {float(r.norm()):.3e}"

    # attention
    xa = x.cuda().requires_grad_(True)
    out = blk.att(xa)
    assert out.requires_grad and out.shape == x.shape
    out.backward(gy.cuda())
    xr = x.clone().requires_grad_(True)
    ro = models.vit_attention(sd, "att.", xr, cfg["n_heads"])
    ro.backward(gy)
    check(blk.att, "att.", out, xa.grad, ro, xr)
    # FFN
    xa = x.cuda().requires_grad_(True)
    out = blk.ffn(xa)
    out.backward(gy.cuda())
    xr = x.clone().requires_grad_(True)
    ro = torch.nn.functional.linear(ops.gelu_erf(torch.nn.functional.linear(xr, sd["ffn.layers.0.weight"], sd["ffn.layers.0.bias"])),
                                    sd["ffn.layers.2.weight"], sd["ffn.layers.2.bias"])
    ro.backward(gy)
    check(blk.ffn, "ffn.", out, xa.grad, ro, xr)
    # LayerNorm (sigma + eps)
    xa = x.cuda().requires_grad_(True)
    out = blk.ln_1(xa)
    out.backward(gy.cuda())
    xr = x.clone().requires_grad_(True)
    ro = ops.layernorm_sigma_eps(xr, sd["ln_1.scale"], sd["ln_1.shift"])
    ro.backward(gy)
    assert rel_l2(out, ro) < 1e-5 and rel_l2(xa.grad, xr.grad) < 1e-4
    assert rel_l2(blk.ln_1.scale.grad, sd["ln_1.scale"].grad) < 1e-4 and rel_l2(blk.ln_1.shift.grad, sd["ln_1.shift"].grad) < 1e-4
    # GELU (bf16 arithmetic in the kernel), an odd element count takes the padded path
    from llm_quest_amd.multimodal.vision_transformer.vit_transformer_block import GELU

    xo = torch.randn(5, 7, generator=g)
    xa = xo.cuda().requires_grad_(True)
    out = GELU()(xa)
    out.backward(torch.ones_like(out))
    xr = xo.to(BF16).float().requires_grad_(True)
    ro = ops.gelu_erf(xr)
    ro.backward(torch.ones_like(ro))
    assert out.dtype == F32 and rel_l2(out, ro) < 6e-3 and rel_l2(xa.grad, xr.grad) < 6e-3
    # eval / no-grad: nothing is kept, nothing requires grad
    with torch.no_grad():
        assert not blk(x.cuda()).requires_grad


def test_rope_apply_positions_outside_the_table_do_not_read_out_of_bounds():
    """A position >= ctx_len (or negative) is an index error upstream (common/rope.py:225-227).  The kernel has no host sync to raise from, so
    such a row comes out as NaN -- loudly wrong -- and nothing outside the coefficient table is read; rows with valid positions are untouched."""
    from llm_quest_amd.common.rope import RoPE

    cos, sin = RoPE.compute_angles(10_000, 16, 8)
    x = torch.randn(2, 3, 4, 16).to(BF16)
    pid = torch.tensor([[0, 3, 7, 2], [1, 8, -1, 5]])
    y = RoPE.apply(x.cuda(), cos.cuda(), sin.cuda(), pid.cuda()).float().cpu()
    ok = torch.ones(2, 4, dtype=torch.bool)
    ok[1, 1] = ok[1, 2] = False
    assert torch.isnan(y[1, :, 1]).all() and torch.isnan(y[1, :, 2]).all()
    good = RoPE.apply(x, cos, sin, pid.clamp(0, 7)).float()
    assert torch.equal(y.permute(0, 2, 1, 3)[ok], good.permute(0, 2, 1, 3)[ok])


@pytest.mark.gpu
def test_linear_with_gelu_epilogue_is_differentiable():
    """ops.LinearFn with the GELU epilogue (the ViT FFN's first projection, vit_transformer_block.py:70-127, used stand-alone) trains: the
    pre-activation is kept by the dual-output GEMM and the backward goes through gelu'(pre).  Against fp32 autograd of the same math."""
    import torch.nn as nn

    from llm_quest_amd import ops

    torch.manual_seed(5)
    lin = nn.Linear(96, 256, bias=True, dtype=torch.bfloat16).cuda()
    x = torch.randn(3, 40, 96, device="cuda").to(torch.bfloat16).requires_grad_(True)
    y = ops.LinearFn.apply(x, lin, lin.weight, lin.bias.detach().float(), True)
    g = torch.randn_like(y)
    y.backward(g)
    x32, w32, b32 = x.detach().float().requires_grad_(True), lin.weight.detach().float().requires_grad_(True), lin.bias.detach().float()
    y32 = torch.nn.functional.gelu(x32 @ w32.T + b32)
    y32.backward(g.float())
    rel = lambda a, b: float((a.float() - b).norm() / b.norm())
    assert rel(y, y32) < 6e-3 and rel(x.grad, x32.grad) < 1e-2, (rel(y, y32), rel(x.grad, x32.grad))
    wg = lin.weight.grad  # attached to the owner's arena by the weight-gradient GEMM
    assert wg is not None and rel(wg, w32.grad) < 1e-2, rel(wg, w32.grad)
