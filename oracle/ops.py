"""Per-op CPU restatements (PyTorch, functional).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Every function reproduces the reference's rounding points (where it casts, which
tensors are bf16) so that on the CPU it is bit-identical to the reference, and
cites the reference file:line it follows.
"""

import math

import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- norms
def rmsnorm(x, weight, eps=1e-6):
    """PytorchRMSNorm (qwen/qwen3/qwen3_attention.py:19-29): full fp32 cast of the input,
    x * rsqrt(mean(x^2) + eps) * w with w promoted to fp32, result cast back to x.dtype."""
    xf = x.to(torch.float32)
    inv = torch.rsqrt(xf.pow(2).mean(dim=-1, keepdim=True) + eps)
    return (xf * inv * weight.to(torch.float32)).to(x.dtype)


def layernorm_sigma_eps(x, scale, shift, eps=1e-5):
    """ViT/GPT LayerNorm (multimodal/vision_transformer/vit_transformer_block.py:12-31,
    gpt/gpt_transformer_block.py:9-39): (x - mean) / (population_std + eps), eps OUTSIDE the sqrt."""
    sd = torch.std(x, dim=-1, keepdim=True, unbiased=False)
    mu = x.mean(dim=-1, keepdim=True)
    return scale * ((x - mu) / (sd + eps)) + shift


def gelu_erf(x):
    """Exact GELU x*0.5*(1+erf(x/sqrt(2))) (vit_transformer_block.py:34-44, gpt_transformer_block.py:42-60)."""
    return x * 0.5 * (1 + torch.erf(x / math.sqrt(2)))


# --------------------------------------------------------------------------- RoPE
def rope_tables(base, head_dim, ctx_len):
    """RoPE.compute_angles without YaRN/partial rotation (common/rope.py:97-168):
    theta_i = base^(-2i/d), angles = outer(pos, theta), layout cat([a, a]) (half-split), fp32."""
    theta = 1.0 / base ** (2 * torch.arange(0, head_dim // 2, dtype=torch.float32) / head_dim)
    pos = torch.arange(0, ctx_len, dtype=torch.float32)
    ang = torch.outer(pos, theta)
    ang = torch.cat([ang, ang], dim=-1)
    return torch.cos(ang), torch.sin(ang)


def rope_apply(x, cos, sin, position_ids=None):
    """RoPE.apply / rotate_half (common/rope.py:171-243): cos/sin are gathered (or sliced to S),
    CAST TO x.dtype FIRST, then cos*x + sin*cat(-x2, x1).  x is (B, H, S, Dh)."""
    s = x.shape[2]
    if position_ids is not None:
        c = cos[position_ids].unsqueeze(1).to(x.dtype)
        sn = sin[position_ids].unsqueeze(1).to(x.dtype)
    else:
        c = cos[:s].to(x.dtype)
        sn = sin[:s].to(x.dtype)
    half = x.shape[-1] // 2
    rot = torch.cat((-x[..., half:], x[..., :half]), dim=-1)
    return c * x + sn * rot


# --------------------------------------------------------------------------- attention
def causal_mask(ctx_len):
    """GlobalBuffers.get_causal_mask (common/buffers.py:25-37): True = masked (strict upper triangle)."""
    return torch.triu(torch.ones(ctx_len, ctx_len, dtype=torch.bool), diagonal=1)


def gqa_attention_core(q, k, v, n_rep, key_mask=None, causal=True):
    """Score/softmax/PV part of GroupedQueryAttention.forward (qwen/qwen3/qwen3_attention.py:121-144).

    q (B,Hq,S,D), k/v (B,Hkv,S,D) in the model dtype.  K/V heads are expanded so that q-head h uses
    kv-head h // n_rep; scores = q @ k^T (tensor in q.dtype), * D**-0.5, masked positions are FILLED with
    finfo(dtype).min/2 (not -inf) after scaling, softmax over keys, @ v.  Returns (B,Hq,S,D).
    """
    hq = q.shape[1]
    head_src = torch.arange(hq) // n_rep
    kx = k[:, head_src]
    vx = v[:, head_src]
    s = q.shape[2]
    scores = (q @ kx.mT) * (q.shape[-1] ** -0.5)
    if causal or key_mask is not None:
        blocked = causal_mask(s) if causal else torch.zeros(s, s, dtype=torch.bool)
        if key_mask is not None:
            blocked = blocked[None, None] | ~key_mask[:, None, None, :].to(torch.bool)
        scores = scores.masked_fill(blocked, torch.finfo(scores.dtype).min / 2)
    w = F.softmax(scores, dim=-1)
    return w @ vx


def full_attention_core(q, k, v):
    """ViTMultiHeadAttention core (multimodal/vision_transformer/vit_attention.py:74-82): no mask."""
    scores = (q @ k.mT) * (q.shape[-1] ** -0.5)
    return torch.softmax(scores, dim=-1) @ v


# --------------------------------------------------------------------------- FFN / adapter
def swiglu_ffn(x, w1, wg, w2):
    """Qwen3 FFN (qwen/qwen3/qwen3_transformer_block.py:48-53): lin2(lin1(x) * silu(lin_gate(x))), no bias."""
    return F.linear(F.linear(x, w1) * F.silu(F.linear(x, wg)), w2)


# --------------------------------------------------------------------------- patch embedding
def patch_embed(img, conv_w, conv_b, cls_token, patch):
    """PatchEmbedding2D.forward (multimodal/vision_transformer/vit_model.py:62-89) written as an explicit
    im2row + GEMM: rows = patches row-major over (ph, pw), K ordered (c, i, j); CLS row prepended."""
    b, c, h, w = img.shape
    gh, gw = h // patch, w // patch
    rows = (
        img.reshape(b, c, gh, patch, gw, patch).permute(0, 2, 4, 1, 3, 5).reshape(b, gh * gw, c * patch * patch)
    )
    proj = rows @ conv_w.reshape(conv_w.shape[0], -1).t() + conv_b
    return torch.cat([cls_token.expand(b, -1, -1), proj], dim=1)


# --------------------------------------------------------------------------- losses
def vlm_loss(logits, labels, text_mask, n_vision):
    """vlm_loss (multimodal/vlm_engine.py:23-41): logits[:, n_vision-1:-1] vs labels with pads -> -100,
    mean CE over kept positions, computed in the logits dtype."""
    shifted = logits[:, n_vision - 1 : -1, :]
    tgt = labels.masked_fill(text_mask == 0, -100)
    return F.cross_entropy(shifted.flatten(0, 1), tgt.flatten(), ignore_index=-100)


def lm_loss(logits, targets):
    """global_loss for a dense model (engine.py:50-72): CE(logits.flatten(0,1), y.flatten()); no MoE term."""
    return F.cross_entropy(logits.flatten(0, 1), targets.flatten())


def lr_at_step(step, total_steps, init_lr, peak_lr, warmup_steps=0, min_lr=None, decay=None):
    """LearningRateScheduler.step (engine.py:114-202) as a pure function of the step index."""
    init = init_lr if warmup_steps > 0 else peak_lr
    if step < warmup_steps:
        return init + (peak_lr - init) / warmup_steps * step
    if min_lr is not None and decay == "cosine":
        k = step - warmup_steps
        kk = total_steps - warmup_steps
        return min_lr + (peak_lr - min_lr) * 0.5 * (1 + math.cos(math.pi * k / kk))
    return peak_lr
