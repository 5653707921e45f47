"""Qwen3-Next attention pieces on HIP kernels -- API of ``llm_quest/qwen/qwen3_next/qwen3_next_attention.py`` as far as the
Qwen3.5 text stack uses it (BASELINE config 5): ``ZeroCenteredRMSNorm``, ``l2_norm``,
``compute_alpha_factor``, ``gated_delta_rule`` and ``GatedAttention``.  Same constructor keys, attribute names and
``state_dict`` keys as upstream; GPU tensors only (no CPU fallback)."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels_q35 as Q
from llm_quest_amd import ops_q35


class ZeroCenteredRMSNorm(nn.Module):
    """x * rsqrt(mean x^2 + eps) * (1 + scale) in fp32, cast back (reference :20-46); ``scale`` starts at zero."""

    def __init__(self, emb_dim, eps=1e-6, dtype=None):
        super().__init__()
        self.scale = nn.Parameter(torch.zeros(emb_dim, dtype=dtype))
        self.eps = eps

    def forward(self, x):
        L.require_gpu(x)
        if x.dtype != torch.bfloat16 or self.scale.dtype != torch.bfloat16:
            raise TypeError("ZeroCenteredRMSNorm runs on bf16 activations and a bf16 scale (the Qwen3.5 configs)")
        return ops_q35.ZCRMSNormFn.apply(x, self, self.scale)


def l2_norm(x):
    """x / max(||x||, 1e-6) over the last dim (reference :51-60).  bf16 (..., d) on the GPU; forward only (inside the model the
    normalisation and its backward are part of the GDN block node)."""
    L.require_gpu(x)
    shp = x.shape
    x2 = x.reshape(-1, shp[-1]).contiguous()
    return Q.l2norm_fwd(x2, 1, shp[-1]).view(shp)


def compute_alpha_factor(log_A, a, dt_bias):
    """exp(-exp(log_A) * softplus(a + dt_bias)) (reference :71-100): log_A fp32 (h,), a bf16 (..., h), dt_bias bf16 (h,) -> fp32.
    Forward only, see ``l2_norm``."""
    L.require_gpu(log_A, a, dt_bias)
    shp = a.shape
    a2 = a.reshape(-1, shp[-1]).contiguous()
    _, alpha = Q.gdn_gates_fwd(a2, a2, log_A, dt_bias)
    return alpha.view(shp)


def gated_delta_rule(queries, keys, values, beta, alpha, prev_state=None):
    """Reference :103-159.  (b, h, s, d) operands, beta / alpha (b, h, s); returns (attn_output, last_state).  Differentiable."""
    L.require_gpu(queries, keys, values, beta, alpha)
    if prev_state is not None:
        # a carried-in recurrent state (b, h, v_head_dim, qk_head_dim): prefill continuation / decode, or training through it (the backward takes the
        # gradient arriving at the returned state and leaves d(prev_state)); the returned state is a new tensor, prev_state is left as it was (as upstream)
        L.require_gpu(prev_state)
        if torch.is_grad_enabled() and any(t.requires_grad for t in (queries, keys, values, beta, alpha, prev_state)):
            return ops_q35.GatedDeltaRuleFn.apply(queries, keys, values, beta, alpha, prev_state)
        b, h, s, dk = queries.shape
        dv = values.shape[-1]
        tm = lambda t: t.permute(0, 2, 1, 3).reshape(b * s, -1).contiguous()
        state = prev_state.to(torch.float32).contiguous().clone()
        o, _, fin = Q.gated_delta_rule_fwd(tm(queries.to(torch.bfloat16)), tm(keys.to(torch.bfloat16)), tm(values.to(torch.bfloat16)),
                                           beta.to(torch.float32).permute(0, 2, 1).reshape(b * s, h).contiguous(),
                                           alpha.to(torch.float32).permute(0, 2, 1).reshape(b * s, h).contiguous(), b, s, h, h, dk, dv, keep=False, state=state)
        return o.view(b, s, h, dv).permute(0, 2, 1, 3).to(queries.dtype), fin
    return ops_q35.GatedDeltaRuleFn.apply(queries, keys, values, beta, alpha)


class GatedAttention(nn.Module):
    """GQA with a sigmoid output gate, zero-centred QK-norm and partial RoPE (reference :162-261)."""

    is_linear = False

    def __init__(self, cfg):
        super().__init__()
        self.d_in = cfg["emb_dim"]
        self.num_heads = cfg["n_heads"]
        self.num_kv_groups = cfg["num_kv_groups"]
        assert self.num_heads % self.num_kv_groups == 0, "num_heads must be divisible by num_kv_groups"
        self.head_dim = cfg["head_dim"]
        self.d_out = self.num_heads * self.head_dim
        self.dtype = cfg["dtype"]
        self.num_repeat = self.num_heads // self.num_kv_groups
        # as upstream (:181): dropout_p of the SDPA call whenever the CONFIG says training.  On the HIP path the weights are dropped inside the attention
        # kernels (Philox masks regenerated in the backward, llm_quest_amd/rng.py); built for the causal mask alone -- with a padding mask it raises
        self.p_dropout = float(cfg["p_dropout"]) if cfg["training"] else 0.0
        if not 0.0 <= self.p_dropout < 1.0:
            raise ValueError(f"dropout probability has to be in [0, 1), got {self.p_dropout}")
        # w_queries_gate | w_keys | w_values are adjacent in the block's arena: one projection GEMM
        self.w_queries_gate = nn.Linear(self.d_in, self.d_out * 2, bias=False, dtype=self.dtype)
        self.w_keys = nn.Linear(self.d_in, self.num_kv_groups * self.head_dim, bias=False, dtype=self.dtype)
        self.w_values = nn.Linear(self.d_in, self.num_kv_groups * self.head_dim, bias=False, dtype=self.dtype)
        self.q_norm = ZeroCenteredRMSNorm(self.head_dim, dtype=self.dtype)
        self.k_norm = ZeroCenteredRMSNorm(self.head_dim, dtype=self.dtype)
        self.out_proj = nn.Linear(self.d_out, self.d_in, bias=False, dtype=self.dtype)

    def forward(self, x, mask, cos, sin, attn_mask=None):
        """x (b, s, d_in); ``mask`` (the inverted causal buffer) is implied by indices in the kernel; cos / sin (ctx, rotation_dim)."""
        b, s, _ = x.shape
        rt = ops_q35.make_runtime(b, s, x.device, cos, sin, attn_mask=attn_mask)
        return ops_q35.run_mixer(self, x, rt)
