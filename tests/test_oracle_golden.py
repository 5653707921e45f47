"""Pin the CPU oracle against fixtures produced by the reference itself (oracle/gen_golden.py).

CPU-only (runs under -m "not gpu").  Integer/index work is bit-exact; floating-point functions are expected to be
bit-identical on the CPU too (same ATen ops in the same order), with a tiny tolerance to survive thread-count changes.
"""

import numpy as np
import pytest
import torch

from conftest import sub_dict
from oracle import index_ops, models, ops
from oracle.gen_golden import TINY_GPT, TINY_QWEN, TINY_VIT


def close(a, b, rtol=1e-6, atol=1e-6):
    a, b = a.float(), b.float()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.allclose(a, b, rtol=rtol, atol=atol), float((a - b).abs().max())


# ------------------------------------------------------------------ index / bit-exact
def test_patch_gather_index(golden):
    t = golden("index_ops")
    for name, (hw, p) in {"p32_4": (32, 4), "p224_16": (224, 16)}.items():
        ref = t[f"patch_gather.{name}"].numpy().astype(np.int64)
        assert np.array_equal(index_ops.patch_gather_index(3, hw, hw, p), ref)


def test_early_fusion_and_masks(golden):
    t = golden("index_ops")
    vis, txt = t["fusion.vis"], t["fusion.txt"]
    src = index_ops.early_fusion_row_source(2, 3, 5)
    out = torch.empty(2, 8, 4)
    for b in range(2):
        for s in range(8):
            which, r = src[b, s]
            out[b, s] = (vis if which == 0 else txt)[b, r]
    assert torch.equal(out, t["fusion.cat"])
    assert np.array_equal(index_ops.fused_attention_mask(t["fusion.text_mask"].numpy(), 3), t["fusion.mask"].numpy().astype(bool))
    labels, rows = index_ops.vlm_label_rows(t["vlm_loss.ids"].numpy(), t["fusion.text_mask"].numpy(), 3)
    assert np.array_equal(labels, t["vlm_loss.labels"].numpy())
    assert rows.tolist() == [2, 3, 4, 5, 6]
    assert np.array_equal(index_ops.gqa_head_map(6, 3), t["gqa.repeat_interleave"].numpy())
    assert torch.equal(ops.causal_mask(8), t["causal_mask.8"].bool())
    vis_m = index_ops.attention_visibility(8)
    assert np.array_equal(~vis_m[0], t["causal_mask.8"].numpy().astype(bool))


def test_vlm_loss(golden):
    t = golden("index_ops")
    got = ops.vlm_loss(t["vlm_loss.logits"], t["vlm_loss.ids"], t["fusion.text_mask"].bool(), 3)
    close(got, t["vlm_loss.loss"])


# ------------------------------------------------------------------ per-op numeric
def test_rmsnorm(golden):
    t = golden("per_op")
    for width in (1024, 128):
        x = t[f"rmsnorm.{width}.x"].clone().requires_grad_(True)
        w = t[f"rmsnorm.{width}.w"].clone().requires_grad_(True)
        y = ops.rmsnorm(x, w)
        assert torch.equal(y, t[f"rmsnorm.{width}.y"])
        y.backward(t[f"rmsnorm.{width}.gy"])
        close(x.grad, t[f"rmsnorm.{width}.gx"], rtol=2e-2, atol=2e-2)
        close(w.grad, t[f"rmsnorm.{width}.gw"], rtol=2e-2, atol=2e-2)


def test_rope(golden):
    t = golden("per_op")
    cos, sin = ops.rope_tables(1_000_000, 128, 96)
    assert torch.equal(cos, t["rope.cos"]) and torch.equal(sin, t["rope.sin"])
    assert torch.equal(ops.rope_apply(t["rope.x"], cos, sin), t["rope.y"])
    assert torch.equal(ops.rope_apply(t["rope.x"], cos, sin, t["rope.pid"]), t["rope.y_pid"])


def test_layernorm_gelu(golden):
    t = golden("per_op")
    x = t["layernorm.x"].clone().requires_grad_(True)
    sc = t["layernorm.scale"].clone().requires_grad_(True)
    sh = t["layernorm.shift"].clone().requires_grad_(True)
    y = ops.layernorm_sigma_eps(x, sc, sh)
    close(y, t["layernorm.y"])
    y.backward(t["layernorm.gy"])
    close(x.grad, t["layernorm.gx"], atol=1e-5)
    close(sc.grad, t["layernorm.gscale"], atol=1e-5)
    close(sh.grad, t["layernorm.gshift"], atol=1e-5)
    close(ops.gelu_erf(t["gelu.x"]), t["gelu.y"])


def test_swiglu(golden):
    t = golden("per_op")
    assert torch.equal(ops.swiglu_ffn(t["swiglu.x"], t["swiglu.w1"], t["swiglu.wg"], t["swiglu.w2"]), t["swiglu.y"])


def test_gqa_module(golden):
    t = golden("per_op")
    sd = sub_dict(t, "gqa.sd.")
    cfg = dict(n_heads=4, num_kv_groups=2, head_dim=128)
    cos, sin = ops.rope_tables(1_000_000, 128, 96)
    y = models.qwen3_attention(sd, "", t["gqa.x"], cfg, cos, sin, key_mask=t["gqa.key_mask"].bool())
    assert torch.equal(y, t["gqa.y"])
    y2 = models.qwen3_attention(sd, "", t["gqa.x"], cfg, cos, sin)
    assert torch.equal(y2, t["gqa.y_nomask"])


def test_gqa_upstream_known_answer(golden):
    """The seeded __main__ demo of qwen3_attention.py:154-185; SURVEY.md section 4 quotes its first output row."""
    t = golden("per_op")
    sd = sub_dict(t, "gqa_toy.sd.")
    cfg = dict(n_heads=6, num_kv_groups=2, head_dim=2)
    cos, sin = ops.rope_tables(10_000, 2, 6)
    y = models.qwen3_attention(sd, "", t["gqa_toy.x"], cfg, cos, sin)
    close(y, t["gqa_toy.y"])
    quoted = torch.tensor([-0.0407, 0.1041, -0.0312, 0.0477, 0.0736])
    assert torch.allclose(y[0, 0], quoted, atol=1e-4)


def test_cross_entropy_ignore_index(golden):
    t = golden("per_op")
    lg = t["ce.logits"].clone().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lg, t["ce.targets"], ignore_index=-100)
    assert torch.equal(loss, t["ce.loss"])
    loss.backward()
    assert torch.equal(lg.grad, t["ce.glogits"])


def test_lr_schedule(golden):
    t = golden("per_op")
    got = [ops.lr_at_step(s, 10, 1e-5, 1e-3, warmup_steps=3, min_lr=1e-4, decay="cosine") for s in range(10)]
    assert np.allclose(np.array(got), t["lr.trace"].numpy(), rtol=1e-12, atol=0)


# ------------------------------------------------------------------ models
def test_qwen3_tiny(golden):
    t = golden("qwen3_tiny")
    sd = {k: v.clone().requires_grad_(v.is_floating_point() and k not in ("cos", "sin")) for k, v in sub_dict(t, "sd.").items()}
    sd["out_head.weight"] = sd["emb_dict.weight"]  # tied (qwen3_model.py:41)
    km = t["in.key_mask"].bool()
    logits = models.qwen3_forward(sd, TINY_QWEN, t["in.ids"], key_mask=km)
    assert torch.equal(logits, t["out.logits"])
    loss = ops.lm_loss(logits, t["in.targets"])
    assert torch.equal(loss, t["out.loss"])
    loss.backward()
    for k, g in sub_dict(t, "grad.").items():
        if k == "out_head.weight":
            continue
        close(sd[k].grad, g, rtol=1e-2, atol=1e-3)
    with torch.no_grad():
        assert torch.equal(models.qwen3_forward(sd, TINY_QWEN, t["in.ids"]), t["out.logits_nomask"])
        # fp32 twin of the same weights
        sd32 = {k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()}
        cfg32 = dict(TINY_QWEN, dtype=torch.float32)
        lg32 = models.qwen3_forward(sd32, cfg32, t["in.ids"], key_mask=km)
        close(lg32, t["twin.logits"], atol=1e-5)


def test_vit_tiny(golden):
    t = golden("vit_tiny")
    sd = {k: v.clone().requires_grad_(True) for k, v in sub_dict(t, "sd.").items()}
    logits = models.vit_forward(sd, TINY_VIT, t["in.image"])
    close(logits, t["out.logits"], atol=1e-5)
    loss = torch.nn.functional.cross_entropy(logits, t["in.labels"])
    close(loss, t["out.loss"], atol=1e-5)
    loss.backward()
    for k, g in sub_dict(t, "grad.").items():
        close(sd[k].grad, g, rtol=1e-4, atol=1e-5)
    with torch.no_grad():
        close(models.vit_forward(sd, TINY_VIT, t["in.image"], output_hidden_states=True), t["out.hidden"], atol=1e-5)


def test_vlm_tiny(golden):
    t = golden("vlm_tiny")
    vit_sd = sub_dict(t, "vit.")
    ad_sd = {k: v.clone().requires_grad_(True) for k, v in sub_dict(t, "ad.").items()}
    llm_sd = {k: v.clone().requires_grad_(v.is_floating_point() and k not in ("cos", "sin")) for k, v in sub_dict(t, "llm.").items()}
    llm_sd["out_head.weight"] = llm_sd["emb_dict.weight"]
    loss, logits, fused = models.vlm_forward_loss(
        vit_sd, TINY_VIT, ad_sd, llm_sd, TINY_QWEN, t["in.image"], t["in.ids"], t["in.text_mask"].bool()
    )
    assert torch.equal(fused, t["out.fused"])
    assert torch.equal(logits, t["out.logits"])
    assert torch.equal(loss, t["out.loss"])
    loss.backward()
    for k, g in sub_dict(t, "grad.llm.").items():
        if k == "out_head.weight":
            continue
        close(llm_sd[k].grad, g, rtol=1e-2, atol=1e-3)
    for k, g in sub_dict(t, "grad.ad.").items():
        close(ad_sd[k].grad, g, rtol=1e-2, atol=1e-3)
    with torch.no_grad():
        out = models.adapter_forward({"adapter.weight": t["ad_simple.weight"]}, t["out.vit_hidden"].to(torch.bfloat16))
        assert torch.equal(out, t["ad_simple.out"])


def test_gpt2_tiny(golden):
    t = golden("gpt2_tiny")
    sd = sub_dict(t, "sd.")
    with torch.no_grad():
        close(models.gpt2_forward(sd, TINY_GPT, t["in.ids"]), t["out.logits"], atol=1e-5)
        close(models.gpt2_forward(sd, TINY_GPT, t["in.ids"], key_mask=t["in.key_mask"].bool()), t["out.logits_masked"], atol=1e-5)
        close(t["out.logits_embedded"], t["out.logits"], atol=1e-5)


# ------------------------------------------------------------------ Qwen3.5 (config 5) vision side + wrapper index ops
def test_qwen35_index_ops(golden):
    from oracle import qwen3_5 as q35

    t = golden("qwen35_vision_tiny")
    assert np.array_equal(q35.patch3d_gather_index(3, 4, 16, 16, 4, 2), t["patch3d.gather"].numpy().astype(np.int64))
    assert np.array_equal(q35.merge_row_source(2, 4, 6, 2), t["merge.rows"].numpy().astype(np.int64))
    cos, sin = q35.vision_rope_tables_2d(10_000, 64, 4, 4)
    assert torch.equal(cos, t["vrope.cos"]) and torch.equal(sin, t["vrope.sin"])
    ids = t["pos3d.ids"].numpy()
    assert np.array_equal(q35.position_ids_3d(ids, [[2, 4, 4]], 999, 2), t["pos3d.out"].numpy())
    assert np.array_equal(q35.position_ids_3d(ids, None, 999, 2), t["pos3d.text_only"].numpy())
    fused = q35.masked_scatter_rows(t["scatter.emb"], t["pos3d.ids"] == 999, t["scatter.vis"])
    assert torch.equal(fused, t["scatter.out"])


def test_qwen35_vision_tower(golden):
    from oracle import qwen3_5 as q35
    from oracle.gen_golden import TINY_Q35_VISION

    t = golden("qwen35_vision_tiny")
    sd = {k: v.clone().requires_grad_(True) for k, v in sub_dict(t, "vis.sd.").items()}
    out = q35.vision35_forward(sd, TINY_Q35_VISION, t["vis.in"])
    close(out, t["vis.out"], atol=2e-5)
    out.backward(t["vis.gout"])
    for k, g in sub_dict(t, "vis.grad.").items():
        if k == "pos_embed.weight":  # only the first gh*gw rows are used
            close(sd[k].grad, g, rtol=1e-4, atol=2e-5)
        else:
            close(sd[k].grad, g, rtol=1e-4, atol=2e-5)


# ----------------------------------------------------------------------------- BASELINE config 5: Qwen3.5 text stack (a24)
def test_qwen35_text_index_and_small_ops(golden):
    from oracle import qwen3_5_text as q35t

    t = golden("qwen35_text_tiny")
    for name, sec in {"s11_11_10": [11, 11, 10], "s2_2_2": [2, 2, 2], "s3_3_2": [3, 3, 2]}.items():
        assert np.array_equal(q35t.mrope_axis_of_frequency(sec, sum(sec)), t[f"mrope.axis.{name}"].numpy()), name
    # the pattern SURVEY.md records for [11, 11, 10]: [T, H, W] x 10, then T, H
    assert q35t.mrope_axis_of_frequency([11, 11, 10], 32).tolist() == [0, 1, 2] * 10 + [0, 1]
    c, s = q35t.mrope_coeffs(t["mrope.cos"], t["mrope.sin"], t["mrope.pid"], [3, 3, 2])
    q = t["mrope.q"]
    out = q35t.rope_partial(q, c.unsqueeze(1).to(q.dtype), s.unsqueeze(1).to(q.dtype))
    assert torch.equal(out, t["mrope.out"])
    assert torch.equal(q35t.zc_rmsnorm(t["zc.x"], t["zc.scale"]), t["zc.out"])
    assert torch.equal(q35t.l2_norm(t["l2.x"]), t["l2.out"])
    assert torch.equal(q35t.alpha_factor(t["alpha.log_A"], t["alpha.a"], t["alpha.dt_bias"]), t["alpha.out"])


def test_qwen35_gated_delta_rule(golden):
    from oracle import qwen3_5_text as q35t

    t = golden("qwen35_text_tiny")
    ins = [t["gdr." + n].clone().requires_grad_(True) for n in ("q", "k", "v", "beta", "alpha")]
    o, state = q35t.gated_delta_rule(*ins)
    assert torch.equal(o, t["gdr.out"]) and torch.equal(state, t["gdr.state"])
    (o.float() * t["gdr.gout"]).sum().backward()
    for n, x in zip(("q", "k", "v", "beta", "alpha"), ins):
        assert torch.allclose(x.grad.float(), t["gdr.grad." + n].float(), rtol=1e-5, atol=1e-6), n


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_qwen35_text_tiny(golden, tag):
    from oracle import qwen3_5_text as q35t
    from oracle.gen_golden import TINY_Q35_TEXT

    t = golden("qwen35_text_tiny")
    cfg = dict(TINY_Q35_TEXT)
    sd = {}
    for k, v in sub_dict(t, f"txt.{tag}.sd.").items():
        if k == "mask":
            sd[k] = v.bool()
        elif k == "out_head.weight":
            continue  # tied to emb_dict.weight
        else:
            sd[k] = v.clone().requires_grad_(v.is_floating_point() and k not in ("cos", "sin"))
    ids, am, pid = t["txt.ids"], t["txt.attn_mask"].bool(), t["txt.pid"]
    logits = q35t.text_model_forward(sd, cfg, x=ids, attn_mask=am, position_ids=pid)
    ref = t[f"txt.{tag}.logits"]
    assert logits.dtype == ref.dtype
    tol = 1e-5 if tag == "fp32" else 2e-2  # bf16: same ops, but ATen's fused SDPA / linear kernels may order sums differently
    assert (logits.float() - ref.float()).norm() <= tol * ref.float().norm()
    (logits.float() * t["txt.gout"]).sum().backward()
    grads = sub_dict(t, f"txt.{tag}.grad.")
    assert set(grads) - {"out_head.weight"} <= set(sd)
    for k, g in grads.items():
        if k == "out_head.weight":
            continue
        got = sd[k].grad.float()
        assert (got - g.float()).norm() <= (1e-4 if tag == "fp32" else 4e-2) * g.float().norm() + 1e-6, k
    with torch.no_grad():
        lo = q35t.text_model_forward(sd, cfg, x=ids)
    assert (lo.float() - t[f"txt.{tag}.logits_text_only"].float()).norm() <= tol * t[f"txt.{tag}.logits_text_only"].float().norm()


@pytest.mark.parametrize("tag", ["fp32", "bf16"])
def test_qwen35_vlm_wrapper_tiny(golden, tag):
    """oracle/qwen3_5.py::vlm35_forward against the reference's ``Qwen3_5VLM.forward`` (qwen3_5_vlm_model.py:178-227): position ids exact,
    logits and every gradient (vision tower + text stack) of the bf16 model and of its fp32 weight twin."""
    from oracle import qwen3_5 as q35
    from oracle.gen_golden import TINY_Q35_TEXT, TINY_Q35_VISION

    t = golden("qwen35_vlm_tiny")
    cfg = {**TINY_Q35_TEXT, **TINY_Q35_VISION, "llm_d_in": TINY_Q35_TEXT["emb_dim"], "image_token_id": 250}
    sd = {}
    for k, v in sub_dict(t, "sd.").items():
        if k.endswith("out_head.weight"):
            continue  # tied to emb_dict.weight
        if k.endswith(".mask") or k == "language_model.mask":
            sd[k] = v.bool()
        elif v.is_floating_point() and not k.endswith((".cos", ".sin")) and k not in ("language_model.cos", "language_model.sin"):
            w = v.float() if (tag == "fp32" and v.dtype == torch.bfloat16) else v
            sd[k] = w.clone().requires_grad_(True)
        else:
            sd[k] = v
    logits, pid = q35.vlm35_forward(sd, cfg, t["in.ids"], t["in.pixels"], t["in.attn_mask"].bool())
    assert torch.equal(pid, t["pos3d"])
    ref = t[f"{tag}.logits"]
    assert logits.dtype == ref.dtype
    tol = 2e-5 if tag == "fp32" else 2e-2
    assert (logits.float() - ref.float()).norm() <= tol * ref.float().norm()
    (logits.float() * t["gout"]).sum().backward()
    grads = sub_dict(t, f"{tag}.grad.")
    seen = 0
    for k, g in grads.items():
        if k.endswith("out_head.weight"):
            continue
        got = sd[k].grad
        if got is None:  # rows of the learned position table beyond the grid, parameters the step does not reach
            assert float(g.float().abs().max()) == 0.0, k
            continue
        seen += 1
        assert (got.float() - g.float()).norm() <= (2e-4 if tag == "fp32" else 4e-2) * g.float().norm() + 1e-6, k
    assert seen > 60
