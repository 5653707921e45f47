"""CPU restatement of the Qwen3.5 text stack (BASELINE config 5).  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): functional, state-dict in / tensors out, every function cites the reference lines it follows.
Pinned by tests/golden/qwen35_text_tiny.safetensors (generated from the reference by oracle/gen_golden.py::gen_qwen35_text).
"""

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


# ------------------------------------------------------------------------------------------------- index maps
def mrope_axis_of_frequency(mrope_section, half_dim):
    """axis[j] in {0: T, 1: H, 2: W} whose position id drives rotary frequency j (llm_quest/common/rope.py:283-292).

    T is the default; H takes j = 1, 4, 7, ... < 3*section[1]; W takes j = 2, 5, 8, ... < 3*section[2].  Integer-exact.
    """
    axis = np.zeros(half_dim, dtype=np.int64)
    for dim, offset in ((1, 1), (2, 2)):
        axis[offset : mrope_section[dim] * 3 : 3] = dim
    return axis


def mrope_coeffs(cos, sin, position_ids, mrope_section):
    """(b, s, rotation_dim) cos / sin for 3-axis position ids (rope.py:297-343): gather per axis, interleave, duplicate."""
    half = cos.shape[-1] // 2
    axis = torch.from_numpy(mrope_axis_of_frequency(mrope_section, half))
    ch, sh = cos[:, :half], sin[:, :half]
    pos = position_ids[axis]  # (half, b, s): the position id that drives each frequency
    j = torch.arange(half).view(half, 1, 1)
    mc = ch[pos, j].permute(1, 2, 0)  # (b, s, half)
    ms = sh[pos, j].permute(1, 2, 0)
    return torch.cat([mc, mc], -1), torch.cat([ms, ms], -1)


def rope_partial(x, cos, sin):
    """cos*x + sin*rotate_half(x) on the first rotation_dim features, the rest passes through (rope.py:226-243, 345-358).
    x: (b, h, s, d); cos/sin broadcastable (.., s, rotation_dim), already in x.dtype."""
    rd = cos.shape[-1]
    xr, rest = x[..., :rd], x[..., rd:]
    rot = torch.cat([-xr[..., rd // 2 :], xr[..., : rd // 2]], -1)
    return torch.cat([cos * xr + sin * rot, rest], -1)


# ------------------------------------------------------------------------------------------------- small ops
def zc_rmsnorm(x, scale, eps=1e-6):
    """ZeroCenteredRMSNorm (qwen3_next_attention.py:20-46): fp32, x * rsqrt(mean x^2 + eps) * (1 + scale), back to x.dtype."""
    xf = x.float()
    return (xf * torch.rsqrt(xf.pow(2).mean(-1, keepdim=True) + eps) * (1.0 + scale)).to(x.dtype)


def l2_norm(x):
    """qwen3_next_attention.py:51-60."""
    return x * torch.clamp(torch.linalg.vector_norm(x, dim=-1, ord=2, keepdim=True), min=1e-6).reciprocal()


def alpha_factor(log_A, a, dt_bias):
    """exp(-exp(log_A) * softplus(a + dt_bias)) (qwen3_next_attention.py:71-100)."""
    return torch.exp(-torch.exp(log_A) * F.softplus(a + dt_bias))


def gated_delta_rule(q, k, v, beta, alpha):
    """Sequential gated delta rule, fp32 (qwen3_next_attention.py:103-159).  q,k: (b,h,s,dk); v: (b,h,s,dv); beta, alpha: (b,h,s).
    S_t = a_t S_{t-1} + b_t (v_t - a_t S_{t-1} k_t) k_t^T ;  o_t = S_t (q_t / sqrt(dk))."""
    dt = q.dtype
    q, k, v, beta, alpha = (t.float() for t in (q, k, v, beta, alpha))
    q = q * q.shape[-1] ** -0.5
    b, h, s, dk = k.shape
    state = torch.zeros(b, h, v.shape[-1], dk)
    out = torch.zeros_like(v)
    for t in range(s):
        kt, vt, qt = k[:, :, t], v[:, :, t], q[:, :, t]
        gated = alpha[:, :, t, None, None] * state
        delta = vt - (gated @ kt.unsqueeze(-1)).squeeze(-1)
        state = gated + (beta[:, :, t, None] * delta).unsqueeze(-1) @ kt.unsqueeze(2)
        out[:, :, t] = (state @ qt.unsqueeze(-1)).squeeze(-1)
    return out.to(dt), state


def causal_depthwise_conv_silu(x, w):
    """x: (b, s, c), w: (c, 1, ksize): depthwise Conv1d with left padding ksize-1, cropped to s, then SiLU
    (qwen3_5_text_model.py:76-88, 139-141)."""
    s = x.shape[1]
    y = F.conv1d(x.transpose(1, 2), w, None, padding=w.shape[-1] - 1, groups=w.shape[0])[..., :s]
    return F.silu(y).transpose(1, 2)


# ------------------------------------------------------------------------------------------------- modules
def gated_attention(sd, pre, cfg, x, allow, cos, sin, position_ids, attn_mask, att_mul=None):
    """MRoPEGatedAttention.forward / GatedAttention.forward (qwen3_5_text_model.py:205-265, qwen3_next_attention.py:204-261).
    allow: (ctx, ctx) bool, True = may attend (the model's inverted causal buffer).  ``att_mul`` stands for SDPA's ``dropout_p`` (:245-253)
    with the mask given: a (b, heads, s, s) tensor of 0 or 1 / (1 - p) applied to the softmax weights."""
    b, s, _ = x.shape
    H, G, D = cfg["n_heads"], cfg["num_kv_groups"], cfg["head_dim"]
    qg = F.linear(x, sd[pre + "w_queries_gate.weight"]).view(b, s, H, 2 * D)
    q, gate = qg[..., :D], qg[..., D:]
    gate_out = torch.sigmoid(gate.reshape(b, s, H * D))
    k = F.linear(x, sd[pre + "w_keys.weight"]).view(b, s, G, D).transpose(1, 2)
    v = F.linear(x, sd[pre + "w_values.weight"]).view(b, s, G, D).transpose(1, 2)
    q = zc_rmsnorm(q.transpose(1, 2), sd[pre + "q_norm.scale"])
    k = zc_rmsnorm(k, sd[pre + "k_norm.scale"])
    if position_ids is None:  # text-only path: plain 1-D RoPE on the rotated quarter (rope.py:180-243)
        c, sn = cos[:s].to(x.dtype), sin[:s].to(x.dtype)
    else:
        c, sn = mrope_coeffs(cos, sin, position_ids, cfg["mrope_section"])
        c, sn = c.unsqueeze(1).to(x.dtype), sn.unsqueeze(1).to(x.dtype)
    q, k = rope_partial(q, c, sn), rope_partial(k, c, sn)
    m = allow[:s, :s]
    if attn_mask is not None:  # as upstream: OR of the allow-mask with the inverted padding mask
        m = m.view(1, 1, s, s) | ~attn_mask.bool().view(b, 1, 1, s)
    if att_mul is None:
        ctx = F.scaled_dot_product_attention(q, k, v, attn_mask=m, is_causal=False, dropout_p=0.0, enable_gqa=True)
    else:  # the math SDPA performs, with the dropout multiplier in the place SDPA applies it (after the softmax, before the product with V)
        kk, vv = k.repeat_interleave(H // G, dim=1), v.repeat_interleave(H // G, dim=1)
        sc = (q.float() @ kk.float().mT) * D ** -0.5
        w = torch.softmax(sc.masked_fill(~m, float("-inf")), dim=-1) * att_mul
        ctx = (w @ vv.float()).to(q.dtype)
    ctx = ctx.transpose(1, 2).contiguous().view(b, s, H * D) * gate_out
    return F.linear(ctx, sd[pre + "out_proj.weight"])


def fused_gated_delta_net(sd, pre, cfg, x, attn_mask):
    """FusedGatedDeltaNet.forward without a cache (qwen3_5_text_model.py:94-191)."""
    b, s, _ = x.shape
    Hqk, Hv, Dk, Dv = cfg["linear_num_qk_heads"], cfg["linear_num_value_heads"], cfg["linear_qk_head_dim"], cfg["linear_value_head_dim"]
    if attn_mask is not None:
        x = x * attn_mask.view(b, s, 1).to(x.dtype)
    fused = F.linear(x, sd[pre + "w_qkv.weight"])
    beta = torch.sigmoid(F.linear(x, sd[pre + "w_beta.weight"]).transpose(1, 2))
    alpha = alpha_factor(sd[pre + "log_A"], F.linear(x, sd[pre + "w_alpha.weight"]), sd[pre + "dt_bias"]).transpose(1, 2)
    fused = causal_depthwise_conv_silu(fused, sd[pre + "conv1d.weight"])
    q, k, v = torch.split(fused, [Hqk * Dk, Hqk * Dk, Hv * Dv], dim=-1)
    q = l2_norm(q.reshape(b, s, Hqk, Dk).transpose(1, 2))
    k = l2_norm(k.reshape(b, s, Hqk, Dk).transpose(1, 2))
    v = v.reshape(b, s, Hv, Dv).transpose(1, 2)
    if Hv // Hqk > 1:
        q = q.repeat_interleave(Hv // Hqk, dim=1)
        k = k.repeat_interleave(Hv // Hqk, dim=1)
    ctx, _ = gated_delta_rule(q, k, v, beta, alpha)
    ctx = ops.rmsnorm(ctx.float(), sd[pre + "post_norm.weight"])  # PytorchRMSNorm in fp32
    ctx = ctx.transpose(1, 2).contiguous().view(b, s, Hv * Dv)
    gate = F.silu(F.linear(x, sd[pre + "w_gate.weight"]).float())
    return F.linear((gate * ctx).to(x.dtype), sd[pre + "out_proj.weight"])


def is_linear_layer(cfg, layer_idx):
    """qwen3_5_text_model.py:287-293."""
    return (layer_idx + 1) % cfg["linear_sdpa_ratio"] != 0


def block(sd, pre, cfg, layer_idx, x, allow, cos, sin, position_ids, attn_mask):
    """Qwen3_5TransformerBlock.forward (qwen3_5_text_model.py:298-325)."""
    h = zc_rmsnorm(x, sd[pre + "norm1.scale"])
    if is_linear_layer(cfg, layer_idx):
        h = fused_gated_delta_net(sd, pre + "att.", cfg, h, attn_mask)
    else:
        h = gated_attention(sd, pre + "att.", cfg, h, allow, cos, sin, position_ids, attn_mask)
    x = h + x
    h = zc_rmsnorm(x, sd[pre + "norm2.scale"])
    h = F.linear(F.linear(h, sd[pre + "ffn.lin1.weight"]) * F.silu(F.linear(h, sd[pre + "ffn.lin_gate.weight"])), sd[pre + "ffn.lin2.weight"])
    return h + x


def text_model_forward(sd, cfg, x=None, attn_mask=None, inputs_embs=None, position_ids=None):
    """Qwen3_5TextModel.forward (qwen3_5_text_model.py:388-417); buffers mask / cos / sin come from the state dict."""
    h = inputs_embs if inputs_embs is not None else F.embedding(x, sd["emb_dict.weight"])
    for i in range(cfg["n_layers"]):
        h = block(sd, f"trf_blocks.{i}.", cfg, i, h, sd["mask"], sd["cos"], sd["sin"], position_ids, attn_mask)
    h = zc_rmsnorm(h, sd["final_norm.scale"])
    return F.linear(h, sd["emb_dict.weight"] if cfg["tie_embeddings"] else sd["out_head.weight"])
