"""RoPE tables (host side, built once at model init) -- mirrors ``llm_quest/common/rope.py``'s public static API.

Only what the hot path uses is implemented: ``compute_angles`` without YaRN / NTK scaling (the Qwen3 configs pass
``smooth_scaling_cfg=None``, common/buffers.py:40) and with optional partial rotation.  Applying the rotation is the
job of the fused QK-norm + RoPE HIP kernel (``mi355_qknorm_rope_fwd``); ``RoPE.apply`` on device tensors routes there.
"""

import torch


class RoPE:
    @staticmethod
    def partial_rotation(head_dim, rotation_factor):
        rot = int(head_dim * rotation_factor)
        return rot - (rot % 2)

    @staticmethod
    def compute_angles(base, head_dim, ctx_len, smooth_scaling_cfg=None, ntk_aware_scaling=True, rotation_factor=1.0, dtype=torch.float32):
        """cos/sin tables, fp32, shape (ctx_len, head_dim), half-split layout [a | a] (reference: rope.py:97-168)."""
        if head_dim % 2:
            raise AssertionError("head dim must be divisible by 2 as we have d/2 pairs of angles")
        if dtype != torch.float32:
            raise AssertionError("RoPE tables are built in float32")
        if smooth_scaling_cfg is not None:
            raise NotImplementedError("YaRN / NTK frequency scaling is outside the Qwen3 hot path (SURVEY.md section 2, row 6)")
        if rotation_factor != 1.0:
            head_dim = RoPE.partial_rotation(head_dim, rotation_factor)
        inv_freq = 1.0 / base ** (2 * torch.arange(0, head_dim // 2, dtype=dtype) / head_dim)
        ang = torch.outer(torch.arange(0, ctx_len, dtype=dtype), inv_freq)
        ang = torch.cat([ang, ang], dim=-1)
        return torch.cos(ang), torch.sin(ang)

    @staticmethod
    def rotate_half(x):
        half = x.shape[-1] // 2
        return torch.cat((-x[..., half:], x[..., :half]), dim=-1)

    @staticmethod
    def apply(x, cos, sin, position_ids=None):
        """Stand-alone RoPE on (b, heads, s, head_dim).  Host tensors only (table checks / tests); on the GPU the
        rotation is fused with the QK-norm inside the attention path and never runs as a separate op."""
        if x.is_cuda:
            raise RuntimeError("RoPE.apply is fused into mi355_qknorm_rope_fwd on the GPU path; call the attention module")
        s = x.shape[2]
        if position_ids is not None:
            c, sn = cos[position_ids].unsqueeze(1).to(x.dtype), sin[position_ids].unsqueeze(1).to(x.dtype)
        else:
            c, sn = cos[:s].to(x.dtype), sin[:s].to(x.dtype)
        return c * x + sn * RoPE.rotate_half(x)
