"""ViT multi-head attention on HIP kernels -- API of ``llm_quest/multimodal/vision_transformer/vit_attention.py``."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K

BF16, F32 = torch.bfloat16, torch.float32


def bf16_cached(module, tag, params, rows_dim0=True):
    """bf16 compute copy of fp32 master parameters (concatenated along dim 0), rebuilt only when a parameter changed.
    The masters stay fp32 so ``state_dict()`` matches the reference; the MFMA GEMMs read the bf16 copy."""
    ver = tuple((p.data_ptr(), p._version) for p in params)
    slot = module.__dict__.setdefault("_bf16_cache", {})
    hit = slot.get(tag)
    if hit is not None and hit[0] == ver:
        return hit[1]
    with torch.no_grad():
        parts = [K.cast(p.detach().reshape(p.shape[0], -1).contiguous(), BF16) for p in params]
        val = parts[0] if len(parts) == 1 else torch.cat(parts, dim=0)
    slot[tag] = (ver, val)
    return val


def f32_cat_cached(module, tag, params):
    ver = tuple((p.data_ptr(), p._version) for p in params)
    slot = module.__dict__.setdefault("_bf16_cache", {})
    hit = slot.get(tag)
    if hit is not None and hit[0] == ver:
        return hit[1]
    with torch.no_grad():
        val = torch.cat([p.detach().reshape(-1) for p in params]).contiguous()
    slot[tag] = (ver, val)
    return val


def refuse_training(module, what):
    if torch.is_grad_enabled() and any(p.requires_grad for p in module.parameters()):
        raise NotImplementedError(
            f"{what}: the HIP ViT path is forward-only in this round (frozen / eval ViT as in the VLM step, "
            "vlm_engine.py:80-83). Freeze the ViT (requires_grad=False) or run under torch.no_grad()."
        )


class ViTMultiHeadAttention(nn.Module):
    """Bidirectional MHA, separate q/k/v Linear layers with optional bias (reference: vit_attention.py:8-91)."""

    def __init__(self, d_in, d_out, dropout, num_heads, qkv_bias=False):
        super().__init__()
        if d_out % num_heads != 0:
            raise ValueError("d_out must be divisible by num_heads")
        self.d_out = d_out
        self.num_heads = num_heads
        self.head_dim = d_out // num_heads
        self.att_scaling = self.head_dim**-0.5
        self.w_queries = nn.Linear(d_in, d_out, bias=qkv_bias)
        self.w_keys = nn.Linear(d_in, d_out, bias=qkv_bias)
        self.w_values = nn.Linear(d_in, d_out, bias=qkv_bias)
        self.dropout = nn.Dropout(dropout)
        self.out_proj = nn.Linear(d_out, d_out)

    def context(self, h_bf16, B, S):
        """LayerNormed tokens (bf16 [B*S, d_in]) -> attention context (bf16 [B*S, d_out]) before out_proj."""
        d = self.d_out
        wqkv = bf16_cached(self, "wqkv", [self.w_queries.weight, self.w_keys.weight, self.w_values.weight])
        bqkv = None
        if self.w_queries.bias is not None:
            bqkv = f32_cat_cached(self, "bqkv", [self.w_queries.bias, self.w_keys.bias, self.w_values.bias])
        qkv = K.gemm(L.GEMM_NT, h_bf16, wqkv, bias=bqkv)
        if self.training and self.dropout.p > 0:  # nn.Dropout on the softmax weights (reference :79), inside the kernel
            from llm_quest_amd import rng

            ctx, _ = K.attn_dropout_fwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], B, S, self.num_heads, self.num_heads, self.head_dim,
                                        self.dropout.p, *rng.draw(), causal=False, scale=self.att_scaling)
            return ctx
        ctx, _ = K.attn_fwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], B, S, self.num_heads, self.num_heads, self.head_dim,
                            key_mask=None, causal=False, scale=self.att_scaling)
        return ctx

    def forward(self, x):
        """x (b, s, d_in) fp32 or bf16 -> (b, s, d_out) in x.dtype."""
        L.require_gpu(x)
        refuse_training(self, "ViTMultiHeadAttention")
        B, S, _ = x.shape
        h = x.reshape(B * S, -1)
        h = K.cast(h.contiguous(), BF16) if h.dtype != BF16 else h.contiguous()
        ctx = self.context(h, B, S)
        wo = bf16_cached(self, "wo", [self.out_proj.weight])
        y = K.gemm(L.GEMM_NT, ctx, wo, bias=self.out_proj.bias.detach(), out_dtype=x.dtype)
        return y.view(B, S, self.d_out)
