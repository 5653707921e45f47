"""Generate golden fixtures by running the REFERENCE in the build container.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Run from the repo root:

    PYTHONDONTWRITEBYTECODE=1 python oracle/gen_golden.py [--ref /root/reference] [--out tests/golden]

It imports the reference package (never copied into this repo, never present on the GPU box), builds seeded
tiny models / per-op cases, and stores inputs, state_dicts, outputs and gradients as ``*.safetensors`` (tensor
data only).  ``torch.__version__`` and the thread count are recorded in each file's metadata.
"""

import argparse
import os
import sys

import torch
from safetensors.torch import save_file

SEED = 123


def _save(path, tensors, note):
    flat = {k: v.detach().clone().contiguous() for k, v in tensors.items()}
    meta = {"torch": torch.__version__, "threads": str(torch.get_num_threads()), "note": note}
    save_file(flat, path, metadata=meta)
    kb = os.path.getsize(path) / 1024
    print(f"wrote {path} ({kb:.0f} KiB, {len(flat)} tensors)")


def _sd(model, prefix="sd."):
    return {prefix + k: v for k, v in model.state_dict().items() if k != "mask"}


def _grads(model, prefix="grad."):
    return {prefix + n: p.grad for n, p in model.named_parameters() if p.grad is not None}


TINY_QWEN = dict(
    vocab_size=512, emb_dim=128, n_layers=2, n_heads=4, num_kv_groups=2, head_dim=128, hidden_dim=256,
    context_length=64, rope_base=1_000_000, dtype=torch.bfloat16, tie_embeddings=True,
)
TINY_VIT = dict(
    img_width=32, img_height=32, patch_size=8, num_channels=3, emb_dim=64, n_layers=2, n_heads=1,
    drop_rate=0.0, qkv_bias=True, num_classes=10,
)
TINY_Q35_VISION = dict(
    vision_n_layers=2, vision_emb_dim=128, vision_hidden_dim=256, vision_num_heads=2, llm_d_in=128, in_channels=3, patch_size=8,
    spatial_merge_size=2, temporal_patch_size=2, num_position_embeddings=64, img_width=32, img_height=32, vision_rope_base=10_000,
)
TINY_GPT = dict(vocab_size=256, context_length=32, emb_dim=64, n_heads=2, n_layers=2, drop_rate=0.0, qkv_bias=True)


def gen_index(out):
    from llm_quest.common.buffers import GlobalBuffers
    from llm_quest.multimodal.vision_transformer.vit_model import PatchEmbedding2D
    from llm_quest.multimodal.vlm_engine import vlm_loss

    t = {}
    # patch order: identity projection turns PatchEmbedding2D into a pure gather of the arange image
    for name, (hw, p) in {"p32_4": (32, 4), "p224_16": (224, 16)}.items():
        k = 3 * p * p
        pe = PatchEmbedding2D(hw, hw, p, 3, k)
        with torch.no_grad():
            pe.conv_proj.weight.copy_(torch.eye(k).view(k, 3, p, p))
            pe.conv_proj.bias.zero_()
            pe.cls_token.zero_()
            img = torch.arange(3 * hw * hw, dtype=torch.float32).view(1, 3, hw, hw)
            t[f"patch_gather.{name}"] = pe(img)[0, 1:].to(torch.int32)  # (num_patches, K) source offsets
    # early-fusion cat + mask + label shift (vlm_engine.py:111-119, 36-39)
    vis = torch.arange(2 * 3 * 4, dtype=torch.float32).view(2, 3, 4)
    txt = 1000 + torch.arange(2 * 5 * 4, dtype=torch.float32).view(2, 5, 4)
    t["fusion.vis"], t["fusion.txt"] = vis, txt
    t["fusion.cat"] = torch.cat([vis, txt], dim=1)
    tm = torch.tensor([[1, 1, 1, 0, 0], [1, 1, 1, 1, 1]], dtype=torch.bool)
    t["fusion.text_mask"] = tm.to(torch.uint8)
    t["fusion.mask"] = torch.cat([torch.ones(2, 3, dtype=torch.bool), tm], dim=1).to(torch.uint8)
    torch.manual_seed(SEED)
    logits = torch.randn(2, 8, 11)
    ids = torch.randint(0, 11, (2, 5))
    t["vlm_loss.logits"], t["vlm_loss.ids"] = logits, ids
    t["vlm_loss.loss"] = vlm_loss(logits, ids, tm, 3)
    t["vlm_loss.labels"] = ids.masked_fill(tm == 0, -100)
    # causal mask, GQA head map
    t["causal_mask.8"] = GlobalBuffers.get_causal_mask(8).to(torch.uint8)
    kv = torch.arange(2 * 3 * 1 * 1).view(1, 3, 2, 1).float()  # (b, kv_heads=3, s=2, d=1)
    t["gqa.repeat_interleave"] = kv.repeat_interleave(2, dim=1)[0, :, 0, 0].to(torch.int32) // 2  # kv head per q head
    _save(os.path.join(out, "index_ops.safetensors"), t, "bit-exact index/gather fixtures")


def gen_ops(out):
    from llm_quest.common.buffers import GlobalBuffers
    from llm_quest.common.rope import RoPE
    from llm_quest.engine import LearningRateScheduler
    from llm_quest.multimodal.vision_transformer.vit_transformer_block import GELU, LayerNorm
    from llm_quest.qwen.qwen3.qwen3_attention import GroupedQueryAttention, PytorchRMSNorm
    from llm_quest.qwen.qwen3.qwen3_transformer_block import FFN

    t = {}
    torch.manual_seed(SEED)
    for width in (1024, 128):
        n = PytorchRMSNorm(width, dtype=torch.bfloat16)
        with torch.no_grad():
            n.weight.copy_(1 + 0.1 * torch.randn(width))
        x = (torch.randn(6, width) * 1.7).to(torch.bfloat16).requires_grad_(True)
        y = n(x)
        g = torch.randn_like(y)
        y.backward(g)
        t[f"rmsnorm.{width}.x"], t[f"rmsnorm.{width}.w"], t[f"rmsnorm.{width}.y"] = x, n.weight, y
        t[f"rmsnorm.{width}.gy"], t[f"rmsnorm.{width}.gx"], t[f"rmsnorm.{width}.gw"] = g, x.grad, n.weight.grad
    # RoPE tables + apply (bf16), with and without position_ids
    cos, sin = RoPE.compute_angles(base=1_000_000, head_dim=128, ctx_len=96)
    t["rope.cos"], t["rope.sin"] = cos, sin
    x = torch.randn(2, 4, 64, 128).to(torch.bfloat16)
    t["rope.x"] = x
    t["rope.y"] = RoPE.apply(x, cos, sin)
    pid = torch.randint(0, 96, (2, 64))
    t["rope.pid"] = pid
    t["rope.y_pid"] = RoPE.apply(x, cos, sin, pid)
    # LayerNorm (sigma + eps), GELU erf
    ln = LayerNorm(768)
    with torch.no_grad():
        ln.scale.copy_(1 + 0.1 * torch.randn(768))
        ln.shift.copy_(0.1 * torch.randn(768))
    x = torch.randn(5, 768, requires_grad=True)
    y = ln(x)
    g = torch.randn_like(y)
    y.backward(g)
    t["layernorm.x"], t["layernorm.scale"], t["layernorm.shift"], t["layernorm.y"] = x, ln.scale, ln.shift, y
    t["layernorm.gy"], t["layernorm.gx"] = g, x.grad
    t["layernorm.gscale"], t["layernorm.gshift"] = ln.scale.grad, ln.shift.grad
    x = torch.randn(4, 256) * 2
    t["gelu.x"], t["gelu.y"] = x, GELU()(x)
    # SwiGLU FFN bf16
    ffn = FFN({"emb_dim": 128, "hidden_dim": 256, "dtype": torch.bfloat16})
    x = torch.randn(3, 7, 128).to(torch.bfloat16)
    t["swiglu.x"], t["swiglu.y"] = x, ffn(x)
    t["swiglu.w1"], t["swiglu.wg"], t["swiglu.w2"] = ffn.lin1.weight, ffn.lin_gate.weight, ffn.lin2.weight
    # GQA module, real head width, with a right-padded key mask, bf16
    att = GroupedQueryAttention(d_in=128, num_heads=4, num_kv_groups=2, head_dim=128, dtype=torch.bfloat16)
    with torch.no_grad():
        att.q_norm.weight.copy_(1 + 0.1 * torch.randn(128))
        att.k_norm.weight.copy_(1 + 0.1 * torch.randn(128))
    x = torch.randn(2, 40, 128).to(torch.bfloat16)
    km = torch.ones(2, 40, dtype=torch.bool)
    km[0, 29:] = False
    mask = GlobalBuffers.get_causal_mask(96)
    t["gqa.x"], t["gqa.key_mask"] = x, km.to(torch.uint8)
    t["gqa.y"] = att(x, mask, cos, sin, km)
    t["gqa.y_nomask"] = att(x, mask, cos, sin)
    for k_, v_ in att.state_dict().items():
        t["gqa.sd." + k_] = v_
    # upstream __main__ known-answer block of qwen3_attention.py:154-185 (seed 123, fp32 toy)
    torch.manual_seed(SEED)
    toy = torch.tensor(
        [[0.43, 0.15, 0.89, 0.15, 0.15], [0.55, 0.87, 0.66, 0.87, 0.87], [0.57, 0.85, 0.64, 0.85, 0.85],
         [0.22, 0.58, 0.33, 0.58, 0.58], [0.77, 0.25, 0.10, 0.25, 0.25], [0.05, 0.80, 0.55, 0.80, 0.80]]
    )
    xb = torch.stack((toy, toy), dim=0)
    m6 = GlobalBuffers.get_causal_mask(6)
    c2, s2 = GlobalBuffers.get_rope_params(6, 10_000, 2)
    toy_att = GroupedQueryAttention(d_in=5, head_dim=2, num_heads=6, num_kv_groups=2)
    t["gqa_toy.x"], t["gqa_toy.y"] = xb, toy_att(xb, m6, c2, s2)
    for k_, v_ in toy_att.state_dict().items():
        t["gqa_toy.sd." + k_] = v_
    # cross-entropy with ignore_index on bf16 logits
    lg = (torch.randn(12, 512) * 3).to(torch.bfloat16).requires_grad_(True)
    tg = torch.randint(0, 512, (12,))
    tg[[2, 7]] = -100
    loss = torch.nn.functional.cross_entropy(lg, tg, ignore_index=-100)
    loss.backward()
    t["ce.logits"], t["ce.targets"], t["ce.loss"], t["ce.glogits"] = lg, tg, loss, lg.grad
    # LR scheduler trace
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=1.0)
    sch = LearningRateScheduler(opt, total_steps=10, init_lr=1e-5, peak_lr=1e-3, warmup_steps=3, min_lr=1e-4, decay="cosine")
    trace = []
    for s in range(10):
        sch.step(s)
        trace.append(sch.current_lr)
    t["lr.trace"] = torch.tensor(trace, dtype=torch.float64)
    _save(os.path.join(out, "per_op.safetensors"), t, "per-op numeric fixtures")


def gen_qwen(out):
    from llm_quest.engine import global_loss
    from llm_quest.qwen.qwen3.qwen3_model import Qwen3Model

    torch.manual_seed(SEED)
    m = Qwen3Model(dict(TINY_QWEN)).train()
    ids = torch.randint(0, 512, (2, 24))
    tgt = torch.randint(0, 512, (2, 24))
    km = torch.ones(2, 24, dtype=torch.bool)
    km[1, 17:] = False
    t = _sd(m)
    t["in.ids"], t["in.targets"], t["in.key_mask"] = ids, tgt, km.to(torch.uint8)
    logits = m(ids, attn_mask=km)
    loss = global_loss(logits, tgt, model=m)
    loss.backward()
    t["out.logits"], t["out.loss"] = logits, loss
    t.update(_grads(m))
    m.zero_grad()
    t["out.logits_nomask"] = m(ids)
    # fp32 twin: same weights upcast
    cfg32 = dict(TINY_QWEN, dtype=torch.float32)
    m32 = Qwen3Model(cfg32).train()
    m32.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in m.state_dict().items()})
    lg32 = m32(ids, attn_mask=km)
    l32 = global_loss(lg32, tgt, model=m32)
    l32.backward()
    t["twin.logits"], t["twin.loss"] = lg32, l32
    t["twin.grad.emb_dict.weight"] = m32.emb_dict.weight.grad
    t["twin.grad.trf_blocks.0.att.w_queries.weight"] = m32.trf_blocks[0].att.w_queries.weight.grad
    t["twin.grad.trf_blocks.1.ffn.lin2.weight"] = m32.trf_blocks[1].ffn.lin2.weight.grad
    _save(os.path.join(out, "qwen3_tiny.safetensors"), t, "tiny Qwen3 dense, bf16 + fp32 twin outputs")


def gen_vit(out):
    from llm_quest.multimodal.vision_transformer.vit_model import ViTModel

    torch.manual_seed(SEED)
    m = ViTModel(dict(TINY_VIT)).train()
    img = torch.randn(3, 3, 32, 32)
    y = torch.randint(0, 10, (3,))
    t = _sd(m)
    t["in.image"], t["in.labels"] = img, y
    logits = m(img)
    loss = torch.nn.functional.cross_entropy(logits, y)
    loss.backward()
    t["out.logits"], t["out.loss"] = logits, loss
    t["out.hidden"] = m(img, output_hidden_states=True)
    t.update(_grads(m))
    _save(os.path.join(out, "vit_tiny.safetensors"), t, "tiny ViT fp32, drop_rate 0")


def gen_vlm(out):
    from llm_quest.multimodal.vision_transformer.vit_engine import ViTAdapter
    from llm_quest.multimodal.vision_transformer.vit_model import ViTModel
    from llm_quest.multimodal.vlm_engine import vlm_loss
    from llm_quest.qwen.qwen3.qwen3_model import Qwen3Model

    torch.manual_seed(SEED)
    vit = ViTModel(dict(TINY_VIT)).eval()
    for p in vit.parameters():
        p.requires_grad = False
    llm = Qwen3Model(dict(TINY_QWEN)).train()
    ad = ViTAdapter(64, 128, adapter_type="ffn", dtype=torch.bfloat16).train()
    img = torch.randn(2, 3, 32, 32)
    ids = torch.randint(0, 512, (2, 20))
    tm = torch.ones(2, 20, dtype=torch.bool)
    tm[0, 13:] = False
    # composition harness of SURVEY.md section 8c over the reference's public attributes
    h = vit(img, output_hidden_states=True)
    ve = ad(h.to(torch.bfloat16))
    te = llm.emb_dict(ids)
    x = torch.cat([ve, te], dim=1)
    nv = ve.shape[1]
    cm = torch.cat([torch.ones(2, nv, dtype=torch.bool), tm], dim=1)
    fused = x
    for blk in llm.trf_blocks:
        x = blk(x, llm.mask, llm.cos, llm.sin, cm, None, None)
    logits = llm.out_head(llm.final_norm(x))
    loss = vlm_loss(logits, ids, tm, nv)
    loss.backward()
    t = {}
    t.update(_sd(vit, "vit."))
    t.update(_sd(llm, "llm."))
    t.update(_sd(ad, "ad."))
    t["in.image"], t["in.ids"], t["in.text_mask"] = img, ids, tm.to(torch.uint8)
    t["out.vit_hidden"], t["out.fused"], t["out.logits"], t["out.loss"] = h, fused, logits, loss
    t.update(_grads(llm, "grad.llm."))
    t.update(_grads(ad, "grad.ad."))
    # simple-adapter variant forward only
    ad2 = ViTAdapter(64, 128, adapter_type="simple", dtype=torch.bfloat16)
    t["ad_simple.weight"] = ad2.adapter.weight
    t["ad_simple.out"] = ad2(h.to(torch.bfloat16))
    # fp32 twin of the whole step (the same bf16 weights upcast, the same inputs): what the bf16 step's gradients are noise around
    llm32 = Qwen3Model(dict(TINY_QWEN, dtype=torch.float32)).train()
    llm32.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in llm.state_dict().items()})
    ad32 = ViTAdapter(64, 128, adapter_type="ffn", dtype=torch.float32).train()
    ad32.load_state_dict({k: v.float() for k, v in ad.state_dict().items()})
    x32 = torch.cat([ad32(h), llm32.emb_dict(ids)], dim=1)
    for blk in llm32.trf_blocks:
        x32 = blk(x32, llm32.mask, llm32.cos, llm32.sin, cm, None, None)
    logits32 = llm32.out_head(llm32.final_norm(x32))
    loss32 = vlm_loss(logits32, ids, tm, nv)
    loss32.backward()
    t["twin.logits"], t["twin.loss"] = logits32, loss32
    t.update(_grads(llm32, "twin.grad.llm."))
    t.update(_grads(ad32, "twin.grad.ad."))
    _save(os.path.join(out, "vlm_tiny.safetensors"), t, "composed ViT+adapter+Qwen3 early-fusion step (config 4 harness)")


def gen_gpt(out):
    from llm_quest.gpt.gpt_model import GPTModel

    torch.manual_seed(SEED)
    m = GPTModel(dict(TINY_GPT)).eval()
    ids = torch.randint(0, 256, (4, 16))
    km = torch.ones(4, 16, dtype=torch.bool)
    km[2, 11:] = False
    t = _sd(m)
    t = {k: v for k, v in t.items() if not k.endswith(".mask")}
    t["in.ids"], t["in.key_mask"] = ids, km.to(torch.uint8)
    with torch.no_grad():
        t["out.logits"] = m(ids)
        t["out.logits_masked"] = m(ids, attn_mask=km)
        emb = m.emb_dict(ids) + m.pos_emb_dict(torch.arange(16))
        t["out.logits_embedded"] = m(emb, input_embedded=True)
    _save(os.path.join(out, "gpt2_tiny.safetensors"), t, "tiny GPT-2 forward (config-1 plumbing)")


def gen_qwen35(out):
    import types

    from llm_quest.common.rope import VisionRoPE
    from llm_quest.qwen.qwen3_5.qwen3_5_vision_model import PatchEmbedding3D, Qwen3_5VisionModel, ViTMergeAdapter
    from llm_quest.qwen.qwen3_5.qwen3_5_vlm_model import Qwen3_5VLM

    t = {}
    torch.manual_seed(SEED)
    # --- index fixtures: Conv3d patch order via an identity projection on an arange clip
    c, frames, hw, p, tp = 3, 4, 16, 4, 2
    k = c * tp * p * p
    pe = PatchEmbedding3D(hw, hw, c, k, p, tp)
    with torch.no_grad():
        pe.conv_proj.weight.copy_(torch.eye(k).view(k, c, tp, p, p))
        pe.conv_proj.bias.zero_()
        clip = torch.arange(c * frames * hw * hw, dtype=torch.float32).view(1, c, frames, hw, hw)
        t["patch3d.gather"] = pe(clip)[0].to(torch.int32)
    # --- merge permutation on an arange grid (LayerNorm made an identity)
    ma = ViTMergeAdapter(8, 8, n_height_patches=4, n_width_patches=6, spatial_merge_size=2)
    x = torch.arange(2 * 24, dtype=torch.float32).view(1, 48, 1).expand(1, 48, 8).contiguous()
    xv = x.view(1, 2, 2, 2, 3, 2, 8).permute(0, 1, 2, 4, 3, 5, 6).contiguous().view(1, -1, 32)
    t["merge.rows"] = xv[0, :, ::8].to(torch.int32)  # source patch row of each of the 4 merged slots
    # --- 2-D RoPE tables
    cos, sin = VisionRoPE.compute_angles_2d(base=10_000, head_dim=64, height_patches=4, width_patches=4)
    t["vrope.cos"], t["vrope.sin"] = cos, sin
    # --- 3-D position ids: 5 text + 8 image (2 frames x 2x2 merged) + 4 text, and a text-only row
    stub = types.SimpleNamespace(image_token_id=999, merge_size=2)
    ids = torch.tensor([[1, 2, 3, 4, 5] + [999] * 8 + [6, 7, 8, 9], [1] * 17])
    feeds = torch.tensor([[2, 4, 4]])
    t["pos3d.ids"] = ids
    t["pos3d.out"] = Qwen3_5VLM.compute_3d_position_ids(stub, ids, feeds)
    t["pos3d.text_only"] = Qwen3_5VLM.compute_3d_position_ids(stub, ids, None).contiguous()
    # --- masked_scatter fusion
    emb = torch.randn(2, 17, 16).to(torch.bfloat16)
    vis = torch.randn(1, 8, 16)
    mask = ids == 999
    t["scatter.emb"], t["scatter.vis"] = emb, vis
    t["scatter.out"] = emb.masked_scatter(mask.unsqueeze(-1).expand_as(emb), vis.to(emb.dtype))
    # --- tiny vision tower: forward + all gradients (fp32, as the reference constructs it)
    vm = Qwen3_5VisionModel(dict(TINY_Q35_VISION)).train()
    with torch.no_grad():  # nn.LayerNorm / biases start at trivial values; perturb them so the test sees them
        for n_, p_ in vm.named_parameters():
            if n_.endswith("bias") or "norm" in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    pix = torch.randn(2, 3, 4, 32, 32)
    outv = vm(pix)
    g = torch.randn_like(outv)
    outv.backward(g)
    for k_, v_ in vm.state_dict().items():
        t["vis.sd." + k_] = v_
    t["vis.in"], t["vis.out"], t["vis.gout"] = pix, outv, g
    for n_, p_ in vm.named_parameters():
        t["vis.grad." + n_] = p_.grad
    _save(os.path.join(out, "qwen35_vision_tiny.safetensors"), t, "Qwen3.5 vision tower (tiny) + wrapper index fixtures")


TINY_Q35_TEXT = dict(
    vocab_size=256, emb_dim=64, hidden_dim=128, n_layers=4, linear_sdpa_ratio=2, n_heads=2, num_kv_groups=1, head_dim=32,
    rope_base=10_000_000, partial_rope_factor=0.5, context_length=64, linear_num_qk_heads=2, linear_num_value_heads=4,
    linear_qk_head_dim=16, linear_value_head_dim=16, linear_conv_kernel_size=4, tie_embeddings=True, p_dropout=0.0, training=False,
    mrope_section=[3, 3, 2],
)


def gen_qwen35_text(out):
    """BASELINE config 5 text stack: per-op vectors + a tiny hybrid model, bf16 and fp32."""
    from llm_quest.common.rope import RoPE
    from llm_quest.qwen.qwen3_5.qwen3_5_text_model import Qwen3_5TextModel
    from llm_quest.qwen.qwen3_next.qwen3_next_attention import ZeroCenteredRMSNorm, compute_alpha_factor, gated_delta_rule, l2_norm

    t = {}
    torch.manual_seed(SEED)
    # --- MRoPE-I axis pattern: per-axis constant tables make the interleave readable as an index map
    for name, sec in {"s11_11_10": [11, 11, 10], "s2_2_2": [2, 2, 2], "s3_3_2": [3, 3, 2]}.items():
        half = sum(sec)
        cos = torch.stack([torch.full((1, 1, half), float(a)) for a in range(3)])
        mc, _ = RoPE.interleave_mrope_coeffs(cos, cos.clone(), sec)
        t[f"mrope.axis.{name}"] = mc[0, 0].to(torch.int32)
    # --- apply_mrope on random q with 3-axis ids and partial rotation
    cos, sin = RoPE.compute_angles(10_000_000, 32, 64, rotation_factor=0.5)
    q = torch.randn(2, 2, 10, 32).to(torch.bfloat16)
    pid = torch.randint(0, 40, (3, 2, 10))
    t["mrope.cos"], t["mrope.sin"], t["mrope.q"], t["mrope.pid"] = cos, sin, q, pid
    t["mrope.out"] = RoPE.apply_mrope(q, cos, sin, pid, [3, 3, 2])
    # --- small ops
    x = torch.randn(3, 5, 64).to(torch.bfloat16)
    norm = ZeroCenteredRMSNorm(64, dtype=torch.bfloat16)
    with torch.no_grad():
        norm.scale.copy_(0.1 * torch.randn(64))
    t["zc.x"], t["zc.scale"], t["zc.out"] = x, norm.scale.detach(), norm(x).detach()
    v = torch.randn(2, 3, 7, 16)
    t["l2.x"], t["l2.out"] = v, l2_norm(v)
    log_A, a, dtb = torch.log(torch.rand(4) * 16), torch.randn(2, 9, 4), torch.ones(4)
    t["alpha.log_A"], t["alpha.a"], t["alpha.dt_bias"], t["alpha.out"] = log_A, a, dtb, compute_alpha_factor(log_A, a, dtb)
    # --- gated delta rule (fp32 recurrence on bf16 operands) with gradients
    qq, kk = (l2_norm(torch.randn(2, 4, 12, 16)).to(torch.bfloat16).requires_grad_(True) for _ in range(2))
    vv = torch.randn(2, 4, 12, 16).to(torch.bfloat16).requires_grad_(True)
    beta = torch.rand(2, 4, 12).requires_grad_(True)
    alpha = (0.5 + 0.5 * torch.rand(2, 4, 12)).requires_grad_(True)
    o, state = gated_delta_rule(qq, kk, vv, beta, alpha)
    go = torch.randn_like(o.float())
    (o.float() * go).sum().backward()
    for n_, ten in (("q", qq), ("k", kk), ("v", vv), ("beta", beta), ("alpha", alpha)):
        t["gdr." + n_], t["gdr.grad." + n_] = ten.detach(), ten.grad
    t["gdr.out"], t["gdr.state"], t["gdr.gout"] = o.detach(), state.detach(), go
    # --- tiny hybrid model: layers 0, 2 FusedGatedDeltaNet, layers 1, 3 MRoPEGatedAttention
    ids = torch.randint(0, 256, (2, 24))
    am = torch.ones(2, 24, dtype=torch.bool)
    am[0, 20:] = False
    pid = torch.arange(24).view(1, 1, 24).repeat(3, 2, 1)
    pid[1, :, 6:14] = 6 + torch.arange(8) // 4  # an "image" span: H / W advance on their own grid
    pid[2, :, 6:14] = 6 + torch.arange(8) % 4
    pid[:, :, 14:] -= 4
    t["txt.ids"], t["txt.attn_mask"], t["txt.pid"] = ids, am.to(torch.uint8), pid
    g = None
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        torch.manual_seed(SEED + 7)
        m = Qwen3_5TextModel({**TINY_Q35_TEXT, "dtype": dt}).train()
        with torch.no_grad():
            for n_, p_ in m.named_parameters():
                if n_.endswith("scale") or n_.endswith("post_norm.weight") or n_.endswith("dt_bias"):
                    p_.add_((0.1 * torch.randn(p_.shape)).to(p_.dtype))
        logits = m(ids, attn_mask=am, position_ids=pid)
        if g is None:
            g = torch.randn(logits.shape)
        (logits.float() * g).sum().backward()
        with torch.no_grad():
            t[f"txt.{tag}.logits_text_only"] = m(ids)
        for k_, v_ in m.state_dict().items():
            t[f"txt.{tag}.sd." + k_] = v_.to(torch.uint8) if v_.dtype == torch.bool else v_
        t[f"txt.{tag}.logits"] = logits.detach()
        for n_, p_ in m.named_parameters():
            t[f"txt.{tag}.grad." + n_] = p_.grad
    t["txt.gout"] = g
    _save(os.path.join(out, "qwen35_text_tiny.safetensors"), t, "Qwen3.5 text stack (tiny hybrid GDN / gated attention) + per-op vectors")


def gen_qwen35_vlm(out):
    """BASELINE config 5 at the wrapper level: the reference's ``Qwen3_5VLM.forward``
    (qwen3_5_vlm_model.py:178-227) on a tiny model -- fp32 vision tower, bf16 hybrid text stack, two samples with 8 image placeholders each
    (2 frame pairs x 2x2 merged rows), a padding mask -- logits and EVERY parameter gradient; plus the same step with the text stack in fp32
    (the bf16 model's weights upcast: a true weight twin, the floor of the 1.5x rule)."""
    from llm_quest.qwen.qwen3_5.qwen3_5_vlm_model import Qwen3_5VLM

    t = {}
    cfg = {**TINY_Q35_TEXT, **TINY_Q35_VISION, "llm_d_in": TINY_Q35_TEXT["emb_dim"], "image_token_id": 250}
    torch.manual_seed(SEED + 17)
    ids = torch.randint(0, 250, (2, 24))
    ids[0, 5:13] = 250
    ids[1, 9:17] = 250
    pix = torch.randn(2, 3, 4, 32, 32)
    am = torch.ones(2, 24, dtype=torch.bool)
    am[1, 21:] = False
    t["in.ids"], t["in.pixels"], t["in.attn_mask"] = ids, pix, am.to(torch.uint8)
    torch.manual_seed(SEED + 19)
    vlm = Qwen3_5VLM({**cfg, "dtype": torch.bfloat16}).train()
    with torch.no_grad():
        for n_, p_ in vlm.named_parameters():
            if n_.endswith("scale") or n_.endswith("post_norm.weight") or n_.endswith("dt_bias") or (n_.startswith("vision_model") and (n_.endswith("bias") or "norm" in n_)):
                p_.add_((0.1 * torch.randn(p_.shape)).to(p_.dtype))
    logits = vlm(ids, image_pixels=pix, attn_mask=am)
    g = torch.randn(logits.shape)
    (logits.float() * g).sum().backward()
    t["gout"] = g
    with torch.no_grad():
        t["pos3d"] = vlm.compute_3d_position_ids(ids, vlm.get_feeds_3d_shape(pix), image_mask=ids == 250)
    t["bf16.logits"] = logits.detach()
    for k_, v_ in vlm.state_dict().items():
        t["sd." + k_] = v_.to(torch.uint8) if v_.dtype == torch.bool else v_
    for n_, p_ in vlm.named_parameters():
        t["bf16.grad." + n_] = p_.grad
    # the fp32 twin: same weights, text stack in fp32
    twin = Qwen3_5VLM({**cfg, "dtype": torch.float32}).train()
    sd = vlm.state_dict()
    twin.load_state_dict({k: (v.float() if v.is_floating_point() else v) for k, v in sd.items()})
    lt = twin(ids, image_pixels=pix, attn_mask=am)
    (lt * g).sum().backward()
    t["fp32.logits"] = lt.detach()
    for n_, p_ in twin.named_parameters():
        t["fp32.grad." + n_] = p_.grad
    _save(os.path.join(out, "qwen35_vlm_tiny.safetensors"), t, "tiny Qwen3_5VLM.forward (fp32 tower + bf16 hybrid text stack): logits + all gradients, fp32 weight twin")


def gen_decode(out):
    """KV-cache decoding: the reference's tiny Qwen3 (weights of qwen3_tiny.safetensors) driven exactly
    as generate_loop_kv_cache does -- prefill with a KVCache, then one-token steps with position_ids -- teacher-forced with seeded
    tokens so every step's logits are comparable, plus the greedy loop itself."""
    from safetensors.torch import load_file

    from llm_quest.generate import generate_loop_kv_cache
    from llm_quest.qwen.qwen3.qwen3_model import Qwen3Model
    from llm_quest.utils import KVCache

    src = load_file(os.path.join(out, "qwen3_tiny.safetensors"))
    sd = {k[3:]: v for k, v in src.items() if k.startswith("sd.")}
    t = {}
    torch.manual_seed(SEED + 11)
    prompt = torch.randint(0, 512, (2, 10))
    forced = torch.randint(0, 512, (6, 2, 1))
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        m = Qwen3Model(dict(TINY_QWEN, dtype=dt)).eval()
        m.load_state_dict({k: (v.to(dt) if v.is_floating_point() and k not in ("cos", "sin") else v) for k, v in sd.items()}, strict=False)
        with torch.inference_mode():
            kv = KVCache(num_layers=2, prompt_len=10, context_len=64)
            t[f"{tag}.prefill"] = m(prompt, kv_cache=kv).clone()
            for i in range(6):
                t[f"{tag}.step{i}"] = m(forced[i], kv_cache=kv, position_ids=torch.tensor([[10 + i]])).clone()
        if tag == "bf16":
            one = prompt[:1]
            t["greedy.ids"] = generate_loop_kv_cache(one, m, max_gen=8, context_length=64, device=torch.device("cpu"))
    t["prompt"], t["forced"] = prompt, forced
    _save(os.path.join(out, "decode_tiny.safetensors"), t, "tiny Qwen3 KV-cache decode: reference logits per step (bf16 + fp32 twin), greedy ids")


def gen_decode35(out):
    """Qwen3_5Cache decoding (second half): the reference's tiny hybrid text model (bf16 weights of
    qwen35_text_tiny.safetensors; fp32 twin = the same weights upcast) driven as qwen3_5_generate_text_only.py does -- prefill with
    MRoPE position ids + cache, then teacher-forced one-token steps."""
    from safetensors.torch import load_file

    from llm_quest.qwen.qwen3_5.qwen3_5_text_model import Qwen3_5TextModel
    from llm_quest.utils import Qwen3_5Cache

    src = load_file(os.path.join(out, "qwen35_text_tiny.safetensors"))
    sd = {k[len("txt.bf16.sd."):]: v for k, v in src.items() if k.startswith("txt.bf16.sd.")}
    sd["mask"] = sd["mask"].bool()
    t = {}
    torch.manual_seed(SEED + 13)
    prompt = torch.randint(0, 256, (2, 9))
    forced = torch.randint(0, 256, (5, 2, 1))
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        m = Qwen3_5TextModel({**TINY_Q35_TEXT, "dtype": dt}).eval()
        keep32 = ("cos", "sin", "mask")
        m.load_state_dict({k: (v if k in keep32 or not v.is_floating_point() or v.dtype == torch.float32 else v.to(dt)) for k, v in sd.items()})
        with torch.inference_mode():
            cache = Qwen3_5Cache(n_layers=4, linear_sdpa_ratio=2, prompt_len=9, context_len=64)
            pid = torch.arange(9).view(1, 1, 9).expand(3, 2, 9)
            t[f"{tag}.prefill"] = m(prompt, position_ids=pid, cache=cache).clone()
            for i in range(5):
                pid = torch.full((3, 2, 1), 9 + i)
                t[f"{tag}.step{i}"] = m(forced[i], position_ids=pid, cache=cache).clone()
    t["prompt"], t["forced"] = prompt, forced
    _save(os.path.join(out, "decode35_tiny.safetensors"), t, "tiny Qwen3.5 text model, Qwen3_5Cache decode: reference logits per step (bf16 + fp32 twin)")


def gen_pipeline(out):
    """Input pipeline: Pillow's bilinear resize -- the third-party arithmetic behind
    transforms.Resize in MultimodalDataset (dataset.py:341-349) -- on seeded random RGB images."""
    import numpy as np
    import PIL
    from PIL import Image

    rng = np.random.default_rng(SEED)
    t = {}
    for name, (h, w, s_) in {"down": (97, 131, 64), "up": (40, 30, 64), "one_pass": (64, 100, 64), "to224": (300, 200, 224)}.items():
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        t[f"resize.{name}.in"] = torch.from_numpy(img)
        t[f"resize.{name}.out"] = torch.from_numpy(np.array(Image.fromarray(img).resize((s_, s_), Image.BILINEAR)))
    from safetensors.torch import save_file

    save_file(t, os.path.join(out, "pipeline.safetensors"), metadata={"what": "Pillow bilinear resize vectors", "pillow": PIL.__version__, "torch": torch.__version__})
    print("wrote pipeline.safetensors", len(t), "tensors")


def gen_signatures(out):
    """Names + argument lists of every symbol on the drop-in boundary, read from the reference's source
    with ast (oracle/signatures.py): data only, no source text."""
    import json

    sys.path.append(os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from oracle import signatures as S

    ref_root = os.path.join(sys.path[0], "llm_quest")
    table = S.collect_tree(ref_root)
    path = os.path.join(out, "signatures.json")
    with open(path, "w") as f:
        json.dump({"note": "reference call signatures (ast, annotations dropped); generated by oracle/gen_golden.py::gen_signatures",
                   "out_of_scope": {m: {k: v for k, v in d.items() if k not in ("*", "keep")} for m, d in S.OUT_OF_SCOPE.items() if d},
                   "modules": table}, f, indent=1, sort_keys=True)
    print(f"wrote {path} ({sum(len(v) for v in table.values())} symbols in {len(table)} modules)")


def _hf_qwen3_tensors(cfg):
    """(name, shape) of a Hugging Face Qwen3 dense checkpoint, from the published architecture (Qwen3ForCausalLM)."""
    d, hd, nq, nkv, ff = cfg["emb_dim"], cfg["head_dim"], cfg["n_heads"], cfg["num_kv_groups"], cfg["hidden_dim"]
    t = [("model.embed_tokens.weight", (cfg["vocab_size"], d)), ("model.norm.weight", (d,)), ("lm_head.weight", (cfg["vocab_size"], d))]
    for i in range(cfg["n_layers"]):
        p = f"model.layers.{i}."
        t += [(p + "self_attn.q_proj.weight", (nq * hd, d)), (p + "self_attn.k_proj.weight", (nkv * hd, d)), (p + "self_attn.v_proj.weight", (nkv * hd, d)),
              (p + "self_attn.o_proj.weight", (d, nq * hd)), (p + "self_attn.q_norm.weight", (hd,)), (p + "self_attn.k_norm.weight", (hd,)),
              (p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
              (p + "mlp.gate_proj.weight", (ff, d)), (p + "mlp.up_proj.weight", (ff, d)), (p + "mlp.down_proj.weight", (d, ff)),
              (p + "self_attn.rotary_emb.inv_freq", (hd // 2,))]  # a tensor older checkpoints carry and no model has: must be reported
    t.append(("model.layers.0.mlp.down_proj.weight_scale", (1,)))  # unknown suffix on a known stem
    return t


def _hf_qwen35_tensors(cfg):
    """(name, shape) of a Hugging Face Qwen3.5 checkpoint (text stack + vision tower + mtp head), from the published architecture."""
    d, hd, nq, nkv, ff = cfg["emb_dim"], cfg["head_dim"], cfg["n_heads"], cfg["num_kv_groups"], cfg["hidden_dim"]
    hk, hv, dk, dv, ks = cfg["linear_num_qk_heads"], cfg["linear_num_value_heads"], cfg["linear_qk_head_dim"], cfg["linear_value_head_dim"], cfg["linear_conv_kernel_size"]
    conv_dim = 2 * hk * dk + hv * dv
    root = "model.language_model."
    t = [(root + "embed_tokens.weight", (cfg["vocab_size"], d)), (root + "norm.weight", (d,)), ("lm_head.weight", (cfg["vocab_size"], d)), ("mtp.fc.weight", (d, 2 * d))]
    for i in range(cfg["n_layers"]):
        p = f"{root}layers.{i}."
        t += [(p + "input_layernorm.weight", (d,)), (p + "post_attention_layernorm.weight", (d,)),
              (p + "mlp.gate_proj.weight", (ff, d)), (p + "mlp.up_proj.weight", (ff, d)), (p + "mlp.down_proj.weight", (d, ff))]
        if (i + 1) % cfg["linear_sdpa_ratio"] == 0:
            t += [(p + "self_attn.q_proj.weight", (2 * nq * hd, d)), (p + "self_attn.k_proj.weight", (nkv * hd, d)), (p + "self_attn.v_proj.weight", (nkv * hd, d)),
                  (p + "self_attn.o_proj.weight", (d, nq * hd)), (p + "self_attn.q_norm.weight", (hd,)), (p + "self_attn.k_norm.weight", (hd,))]
        else:
            t += [(p + "linear_attn.A_log", (hv,)), (p + "linear_attn.dt_bias", (hv,)), (p + "linear_attn.in_proj_qkv.weight", (conv_dim, d)),
                  (p + "linear_attn.in_proj_z.weight", (hv * dv, d)), (p + "linear_attn.in_proj_b.weight", (hv, d)), (p + "linear_attn.in_proj_a.weight", (hv, d)),
                  (p + "linear_attn.conv1d.weight", (conv_dim, 1, ks)), (p + "linear_attn.norm.weight", (dv,)), (p + "linear_attn.out_proj.weight", (d, hv * dv))]
    ve, vh, ps, tp, m = cfg["vision_emb_dim"], cfg["vision_hidden_dim"], cfg["patch_size"], cfg["temporal_patch_size"], cfg["spatial_merge_size"]
    v = "model.visual."
    t += [(v + "patch_embed.proj.weight", (ve, cfg["in_channels"], tp, ps, ps)), (v + "patch_embed.proj.bias", (ve,)), (v + "pos_embed.weight", (cfg["num_position_embeddings"], ve))]
    for i in range(cfg["vision_n_layers"]):
        p = f"{v}blocks.{i}."
        t += [(p + "norm1.weight", (ve,)), (p + "norm1.bias", (ve,)), (p + "norm2.weight", (ve,)), (p + "norm2.bias", (ve,)),
              (p + "attn.qkv.weight", (3 * ve, ve)), (p + "attn.qkv.bias", (3 * ve,)), (p + "attn.proj.weight", (ve, ve)), (p + "attn.proj.bias", (ve,)),
              (p + "mlp.linear_fc1.weight", (vh, ve)), (p + "mlp.linear_fc1.bias", (vh,)), (p + "mlp.linear_fc2.weight", (ve, vh)), (p + "mlp.linear_fc2.bias", (ve,))]
    t += [(v + "merger.norm.weight", (ve,)), (v + "merger.norm.bias", (ve,)), (v + "merger.linear_fc1.weight", (ve * m * m, ve * m * m)),
          (v + "merger.linear_fc1.bias", (ve * m * m,)), (v + "merger.linear_fc2.weight", (cfg["llm_d_in"], ve * m * m)), (v + "merger.linear_fc2.bias", (cfg["llm_d_in"],))]
    t.append((root + "layers.0.linear_attn.dt_bias_extra", (hv,)))  # unknown tensor: must be reported, not loaded
    return t




def gen_weight_maps(out):
    """Row f2: the REFERENCE's ``convert_weights`` + rule tables applied to synthetic Hugging-Face-named checkpoints.  The fixture
    holds the checkpoint's (name, shape) list and, per loader, which of this package's parameter names each tensor reached."""
    import contextlib
    import io
    import json

    from llm_quest.qwen.qwen3 import qwen3_weight_loading as W3
    from llm_quest.qwen.qwen3.qwen3_model import Qwen3Model
    from llm_quest.qwen.qwen3_5 import qwen3_5_weight_loading as W35
    from llm_quest.qwen.qwen3_5.qwen3_5_vlm_model import Qwen3_5VLM
    from llm_quest.utils import convert_weights

    def run(tensors, state, rules, ignored=None):
        hf = {n: torch.zeros(s) for n, s in tensors}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            conv = convert_weights(hf, state, rules, ignored_prefixes=ignored)
        return {"loaded": sorted(conv), "warnings": sum(1 for line in buf.getvalue().splitlines() if line.startswith("WARNING")),
                "skipped": sum(1 for n, _ in tensors if ignored and n.startswith(tuple(ignored)))}

    res = {}
    for tie in (True, False):
        cfg = dict(TINY_QWEN, head_dim=32, tie_embeddings=tie, model_type="dense")
        tensors = _hf_qwen3_tensors(cfg)
        res[f"qwen3_tie{int(tie)}"] = {"cfg": {k: v for k, v in cfg.items() if k != "dtype"}, "tensors": [[n, list(s)] for n, s in tensors],
                                        "result": run(tensors, Qwen3Model(cfg).state_dict(), W3.get_remapping_rules(cfg))}
    cfg = dict(TINY_Q35_TEXT, **TINY_Q35_VISION, image_token_id=250, dtype=torch.bfloat16)
    vlm = Qwen3_5VLM(cfg)
    tensors = _hf_qwen35_tensors(cfg)
    res["qwen35"] = {
        "cfg": {k: v for k, v in cfg.items() if k != "dtype"}, "tensors": [[n, list(s)] for n, s in tensors],
        "text": run(tensors, vlm.language_model.state_dict(), W35.get_remapping_rules(), ("model.visual.", "mtp.")),
        "vision": run(tensors, vlm.vision_model.state_dict(), W35.get_vision_remapping_rules(), ("model.language_model.", "mtp.")),
    }
    path = os.path.join(out, "weight_maps.json")
    with open(path, "w") as f:
        json.dump(res, f, indent=1, sort_keys=True)
    print(f"wrote {path}")


def gen_rope_extra(out):
    """The rest of the RoPE / VisionRoPE boundary (common/rope.py): YaRN / NTK tables, fp32 and partial-rotation apply, 2-D axial
    apply over two frames, and the gradients autograd gives through apply / apply_mrope."""
    from llm_quest.common.rope import RoPE, VisionRoPE

    t = {}
    torch.manual_seed(SEED)
    yarn = dict(factor=4.0, alpha=1.0, beta=32.0, og_ctx_len=64, ctx_len=256)
    for tag, ntk in (("ntk", True), ("plain", False)):
        c, s_ = RoPE.compute_angles(10_000, 64, 256, smooth_scaling_cfg=yarn, ntk_aware_scaling=ntk)
        t[f"yarn.{tag}.cos"], t[f"yarn.{tag}.sin"] = c, s_
    t["yarn.theta"] = RoPE.wavelength_scaling(10_000, 64, yarn)
    # partial rotation (32 of 64 features) with position ids, bf16 and fp32, forward + gradient
    cos, sin = RoPE.compute_angles(10_000, 64, 48, rotation_factor=0.5)
    pid = torch.randint(0, 48, (2, 16))
    t["part.cos"], t["part.sin"], t["part.pid"] = cos, sin, pid
    for tag, dt in (("bf16", torch.bfloat16), ("fp32", torch.float32)):
        x = torch.randn(2, 3, 16, 64).to(dt).requires_grad_(True)
        g = torch.randn(2, 3, 16, 64).to(dt)
        y = RoPE.apply(x, cos, sin, pid)
        y.backward(g)
        t[f"part.{tag}.x"], t[f"part.{tag}.g"], t[f"part.{tag}.y"], t[f"part.{tag}.gx"] = x.detach(), g, y.detach(), x.grad
    # heads stored token-major (b, s, h, d) and handed over as the transposed view, as the attention modules do
    xt = torch.randn(2, 16, 3, 64).to(torch.bfloat16)
    t["tm.x"] = xt
    t["tm.y"] = RoPE.apply(xt.transpose(1, 2), cos, sin).contiguous()
    # MRoPE-I gradient
    mc, ms = RoPE.compute_angles(10_000_000, 32, 64, rotation_factor=0.5)
    q = torch.randn(2, 2, 10, 32).to(torch.bfloat16).requires_grad_(True)
    mp = torch.randint(0, 40, (3, 2, 10))
    gq = torch.randn(2, 2, 10, 32).to(torch.bfloat16)
    yo = RoPE.apply_mrope(q, mc, ms, mp, [3, 3, 2])
    yo.backward(gq)
    t["mrope.cos"], t["mrope.sin"], t["mrope.pid"], t["mrope.q"], t["mrope.g"], t["mrope.y"], t["mrope.gq"] = mc, ms, mp, q.detach(), gq, yo.detach(), q.grad
    # 2-D axial RoPE over two frames
    vc, vs = VisionRoPE.compute_angles_2d(10_000, 64, 3, 5, num_frames=2)
    xv = torch.randn(2, 2, 30, 64).to(torch.bfloat16)
    t["vis.cos"], t["vis.sin"], t["vis.x"], t["vis.y"] = vc, vs, xv, VisionRoPE.apply(xv, vc, vs)
    _save(os.path.join(out, "rope_extra.safetensors"), t, "RoPE / VisionRoPE boundary: YaRN tables, partial / fp32 / token-major / MRoPE / 2-D apply and gradients")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=os.path.join(os.path.dirname(__file__), "..", "tests", "golden"))
    args = ap.parse_args()
    sys.dont_write_bytecode = True
    sys.path.insert(0, args.ref)
    os.makedirs(args.out, exist_ok=True)
    torch.set_num_threads(8)
    only = os.environ.get("GOLDEN_ONLY")
    for fn in (gen_index, gen_ops, gen_qwen, gen_vit, gen_vlm, gen_gpt, gen_qwen35, gen_qwen35_text, gen_qwen35_vlm, gen_pipeline, gen_decode, gen_decode35, gen_signatures, gen_weight_maps, gen_rope_extra):
        if only and fn.__name__ != only:
            continue
        fn(args.out)


if __name__ == "__main__":
    main()
