"""Qwen3 attention modules on HIP kernels -- API of ``llm_quest/qwen/qwen3/qwen3_attention.py``."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd import ops


class PytorchRMSNorm(nn.Module):
    """RMSNorm computed fully in fp32 and cast back (reference: qwen3_attention.py:19-29).  ``weight`` key as upstream."""

    def __init__(self, emb_dim, eps=1e-6, dtype=None):
        super().__init__()
        self.eps = eps
        self.normalized_shape = (emb_dim,)
        self.weight = nn.Parameter(torch.ones(emb_dim, dtype=dtype))

    def forward(self, x):
        L.require_gpu(x)
        return ops.RMSNormFn.apply(x, self, self.weight)


class _AttentionFn(torch.autograd.Function):
    """Stand-alone GroupedQueryAttention (when the module is called outside a TransformerBlock)."""

    @staticmethod
    def forward(ctx, x, att, rt, keep, *params):
        arena = ops.arena_for(att)
        B, S, d = x.shape
        h1 = x.reshape(B * S, d).contiguous()
        c, saved = ops.attention_forward(att, arena, h1, rt)
        y = K.gemm(L.GEMM_NT, c, att.out_proj.weight)
        ctx.att, ctx.rt, ctx.saved, ctx.shape = att, rt, (h1, c, saved) if keep else None, (B, S, d)
        return y.view(B, S, -1)

    @staticmethod
    def backward(ctx, dy):
        att, (B, S, d) = ctx.att, ctx.shape
        arena = ops.arena_for(att)
        h1, c, saved = ctx.saved
        dy2 = dy.reshape(B * S, -1).contiguous()
        dctx = K.dgrad(dy2, att.out_proj.weight)
        ops._wgrad(arena, att.out_proj.weight, None, dy2, c)
        dh1 = ops.attention_backward(att, arena, h1, c, saved, dctx, ctx.rt)
        ctx.saved = None
        return (dh1.view(B, S, d), None, None, None) + (None,) * len(att._param_list)


class GroupedQueryAttention(nn.Module):
    """GQA with QK-RMSNorm before RoPE, no biases (reference: qwen3_attention.py:32-150).

    Parameters keep the reference's names; inside a TransformerBlock they are views into the block's arena so the
    three input projections run as one GEMM.
    """

    def __init__(self, d_in, num_heads, num_kv_groups, head_dim, dtype=None, layer_idx=None):
        super().__init__()
        assert num_heads % num_kv_groups == 0, "num_heads must be divisible by num_kv_groups"
        self.layer_idx = layer_idx
        self.num_heads = num_heads
        self.head_dim = head_dim
        self.d_out = num_heads * head_dim
        self.att_scaling = head_dim**-0.5
        self.num_kv_groups = num_kv_groups
        self.num_repeat = num_heads // num_kv_groups
        # declaration order = arena order: q|k|v must be adjacent for the fused projection
        self.w_queries = nn.Linear(d_in, self.d_out, bias=False, dtype=dtype)
        self.w_keys = nn.Linear(d_in, num_kv_groups * head_dim, bias=False, dtype=dtype)
        self.w_values = nn.Linear(d_in, num_kv_groups * head_dim, bias=False, dtype=dtype)
        self.out_proj = nn.Linear(self.d_out, d_in, bias=False, dtype=dtype)
        self.q_norm = PytorchRMSNorm(head_dim, dtype=dtype)
        self.k_norm = PytorchRMSNorm(head_dim, dtype=dtype)

    def forward(self, x, mask, cos, sin, attn_mask=None, kv_cache=None, position_ids=None):
        """x (b, s, d_in); ``mask`` (the dense causal buffer) is accepted for API parity and never read."""
        L.require_gpu(x)
        B, S, _ = x.shape
        if kv_cache is not None:  # inference (reference :117-118): prefill / one-token decode through utils.KVCache, no autograd
            from llm_quest_amd import ops_decode as OD

            h1 = x.reshape(B * S, -1)
            h1 = (h1 if h1.dtype == torch.bfloat16 else K.cast(h1.contiguous(), torch.bfloat16)).contiguous()
            pos = OD.cached_positions(kv_cache, B, S, x.device, position_ids, reference_default=True)
            km = OD.cached_key_mask(attn_mask, kv_cache, B, S, x.device)
            y = OD.attention_cached(self, h1, B, S, cos, sin, pos, km, kv_cache)
            return (y if y.dtype == x.dtype else K.cast(y, x.dtype)).view(B, S, -1)
        rt = ops.make_runtime(B, S, x.device, cos, sin, attn_mask, position_ids)
        if not hasattr(self, "_param_list"):
            object.__setattr__(self, "_param_list", list(self.parameters()))
        return _AttentionFn.apply(x, self, rt, torch.is_grad_enabled(), *self._param_list)
