"""Host side of the RoPE boundary (common/rope.py, common/buffers.py) against fixtures the reference produced: tables (plain, partial,
YaRN / NTK, 2-D axial), the causal mask, and apply / apply_mrope / VisionRoPE.apply with their autograd gradients -- all bit-exact."""

import torch

from llm_quest_amd.common.buffers import GlobalBuffers
from llm_quest_amd.common.rope import RoPE, VisionRoPE


def test_tables_equal_the_reference(golden):
    t = golden("per_op")
    c, s = RoPE.compute_angles(base=1_000_000, head_dim=128, ctx_len=96)
    assert torch.equal(c, t["rope.cos"]) and torch.equal(s, t["rope.sin"])
    q = golden("qwen3_tiny")
    c, s = GlobalBuffers.get_rope_params(64, 1_000_000, 128)
    assert torch.equal(c, q["sd.cos"]) and torch.equal(s, q["sd.sin"])
    x35 = golden("qwen35_text_tiny")
    c, s = RoPE.compute_angles(10_000_000, 32, 64, rotation_factor=0.5)
    assert c.shape == (64, 16) and torch.equal(c, x35["mrope.cos"]) and torch.equal(s, x35["mrope.sin"])
    v = golden("qwen35_vision_tiny")
    c, s = VisionRoPE.compute_angles_2d(base=10_000, head_dim=64, height_patches=4, width_patches=4)
    assert torch.equal(c, v["vrope.cos"]) and torch.equal(s, v["vrope.sin"])
    e = golden("rope_extra")
    yarn = dict(factor=4.0, alpha=1.0, beta=32.0, og_ctx_len=64, ctx_len=256)
    assert torch.equal(RoPE.wavelength_scaling(10_000, 64, yarn), e["yarn.theta"])
    for tag, ntk in (("ntk", True), ("plain", False)):
        c, s = RoPE.compute_angles(10_000, 64, 256, smooth_scaling_cfg=yarn, ntk_aware_scaling=ntk)
        assert torch.equal(c, e[f"yarn.{tag}.cos"]) and torch.equal(s, e[f"yarn.{tag}.sin"]), tag
    c, s = VisionRoPE.compute_angles_2d(10_000, 64, 3, 5, num_frames=2)
    assert torch.equal(c, e["vis.cos"]) and torch.equal(s, e["vis.sin"])
    assert RoPE.partial_rotation(6, 0.5) == 3  # the reference does not round to an even width here (rope.py:8-30)
    assert torch.equal(GlobalBuffers.get_causal_mask(8), golden("index_ops")["causal_mask.8"].bool())


def test_host_apply_equals_the_reference(golden):
    t, e, x35 = golden("per_op"), golden("rope_extra"), golden("qwen35_text_tiny")
    assert torch.equal(RoPE.apply(t["rope.x"], t["rope.cos"], t["rope.sin"]), t["rope.y"])
    assert torch.equal(RoPE.apply(t["rope.x"], t["rope.cos"], t["rope.sin"], t["rope.pid"]), t["rope.y_pid"])
    for tag in ("bf16", "fp32"):
        x = e[f"part.{tag}.x"].clone().requires_grad_(True)
        y = RoPE.apply(x, e["part.cos"], e["part.sin"], e["part.pid"])
        y.backward(e[f"part.{tag}.g"])
        assert torch.equal(y, e[f"part.{tag}.y"]) and torch.equal(x.grad, e[f"part.{tag}.gx"]), tag
    assert torch.equal(RoPE.apply_mrope(x35["mrope.q"], x35["mrope.cos"], x35["mrope.sin"], x35["mrope.pid"], [3, 3, 2]), x35["mrope.out"])
    for name, sec in {"s11_11_10": [11, 11, 10], "s2_2_2": [2, 2, 2], "s3_3_2": [3, 3, 2]}.items():
        axes = torch.stack([torch.full((1, 1, sum(sec)), float(a)) for a in range(3)])
        mc, ms = RoPE.interleave_mrope_coeffs(axes, axes.clone(), sec)
        assert torch.equal(mc[0, 0].to(torch.int32), x35[f"mrope.axis.{name}"]) and torch.equal(mc, ms)
    assert torch.equal(VisionRoPE.apply(e["vis.x"], e["vis.cos"], e["vis.sin"]), e["vis.y"])
