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


def split3_cached(module, tag, params):
    """[hi | hi | lo] bf16 column blocks of fp32 master weights (concatenated along dim 0): the weight operand of the fp32-grade tower's GEMMs
    (csrc/tower_f32.hip; the activation side is ``kernels.split3`` in the [hi | lo | hi] order).  Rebuilt only when a parameter changed."""
    ver = tuple((p.data_ptr(), p._version) for p in params)
    slot = module.__dict__.setdefault("_bf16_cache", {})
    hit = slot.get(tag)
    if hit is not None and hit[0] == ver:
        return hit[1]
    with torch.no_grad():
        parts = [K.split3(p.detach().reshape(p.shape[0], -1).to(F32).contiguous(), weight_order=True) for p in params]
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

    def context_f32(self, h3, B, S):
        """The same at the reference's fp32 precision (the frozen tower of multimodal/vlm_engine.py:99-104): ``h3`` = split3 of the LayerNormed fp32
        tokens -> fp32 context [B*S, d_out] before out_proj.  fp32-grade projections (three bf16 MFMA products), exact-fp32 attention."""
        d = self.d_out
        wqkv = split3_cached(self, "wqkv3", [self.w_queries.weight, self.w_keys.weight, self.w_values.weight])
        bqkv = None
        if self.w_queries.bias is not None:
            bqkv = f32_cat_cached(self, "bqkv", [self.w_queries.bias, self.w_keys.bias, self.w_values.bias])
        qkv = K.gemm(L.GEMM_NT, h3, wqkv, bias=bqkv, out_dtype=F32)
        return K.attn_f32_fwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], B, S, self.num_heads, self.head_dim, scale=self.att_scaling)

    def forward(self, x):
        """x (b, s, d_in) fp32 or bf16 -> (b, s, d_out) in x.dtype.  An ordinary autograd module as upstream (vit_attention.py:44-91): one
        node over the (forward, backward) pair of ``vit_train``; without grad mode the forward keeps nothing."""
        from llm_quest_amd.multimodal.vision_transformer import vit_train as T

        B, S, _ = x.shape

        def fwd(t):
            y, saved = T.att_forward(self, T.as_bf16_rows(t), B, S, self.training, out_dtype=F32)
            return T.like(y, t, self.d_out), saved

        def bwd(saved, dy):
            return T.like(T.att_backward(self, saved, dy.reshape(B * S, self.d_out)), dy, x.shape[-1])

        return T.run_piece(self, x, fwd, bwd)
