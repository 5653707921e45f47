"""Rotary position embeddings -- public static API of ``llm_quest/common/rope.py`` (``RoPE`` and ``VisionRoPE``).

Tables (``compute_angles``, ``compute_angles_2d``, YaRN / NTK frequency scaling) are host-side fp32 math done once at model
init.  Applying a rotation has two implementations behind one signature:

  * device tensors: the HIP kernel ``mi355_rope_apply`` (csrc/rope_dropout.hip) with the reference's rounding points, wrapped
    in an autograd node whose backward is the kernel's adjoint mode; MRoPE-I first gathers its interleaved per-token coefficient
    rows with ``mi355_mrope_table`` (an exact gather), so 1-D RoPE, partial rotation, 2-D axial RoPE and MRoPE are one kernel;
  * host tensors (table checks, the GPT-2 CPU plumbing of BASELINE config 1): the same arithmetic in torch ops.

Inside the training step the rotation never runs on its own: it is fused with the QK-norm (``mi355_qknorm_rope_fwd``,
``mi355_headnorm_rope_fwd``).  These entry points are the drop-in boundary for callers that use ``RoPE`` directly.
"""

import torch


def _half_turn(x):
    h = x.shape[-1] // 2
    return torch.cat((-x[..., h:], x[..., :h]), dim=-1)


def _host_rotate(x, cos_b, sin_b):
    """x (..., head_dim); cos_b / sin_b broadcastable (..., R) already in x.dtype; features past R pass through."""
    R = cos_b.shape[-1]
    if R < x.shape[-1]:
        head, rest = x[..., :R], x[..., R:]
        return torch.cat((cos_b * head + sin_b * _half_turn(head), rest), dim=-1)
    return cos_b * x + sin_b * _half_turn(x)


class _RopeFn(torch.autograd.Function):
    """out = rope(x) through mi355_rope_apply; backward = the adjoint rotation of the incoming gradient."""

    @staticmethod
    def forward(ctx, x, cos_t, sin_t, idx):
        from llm_quest_amd import kernels as K

        ctx.saved = (cos_t, sin_t, idx)
        return K.rope_apply(x, cos_t, sin_t, idx)

    @staticmethod
    def backward(ctx, dy):
        from llm_quest_amd import kernels as K

        cos_t, sin_t, idx = ctx.saved
        if dy.stride(-1) != 1:
            dy = dy.contiguous()
        return K.rope_apply(dy, cos_t, sin_t, idx, transpose=True), None, None, None


def _device_tables(cos, sin, device):
    c = cos.to(device=device, dtype=torch.float32)
    s = sin.to(device=device, dtype=torch.float32)
    return (c if c.is_contiguous() else c.contiguous()), (s if s.is_contiguous() else s.contiguous())


class RoPE:
    @staticmethod
    def partial_rotation(head_dim, factor):
        """Number of leading features that rotate (reference rope.py:8-30; an odd result loses its last feature in
        ``compute_angles``, which builds head_dim // 2 frequencies)."""
        assert 0 < factor <= 1.0, "rotation factor must be greater than 0 and less than or equal to 1.0"
        return int(head_dim * factor)

    @staticmethod
    def ntk_aware_base_scaling(theta_base, head_dim, ctx_len, old_ctx_len):
        """NTK-aware base for a context extension old_ctx_len -> ctx_len (reference rope.py:32-37)."""
        return theta_base * (ctx_len / old_ctx_len) ** (head_dim / (head_dim - 2))

    @staticmethod
    def wavelength_scaling(base, head_dim, freq_cfg, ntk_aware_scaling=True, dtype=torch.float32):
        """YaRN "NTK by parts" frequencies (reference rope.py:39-95): wavelengths shorter than the original context keep their
        frequency, long ones are divided by ``factor``, the band alpha <= og_ctx_len / wavelength <= beta blends linearly."""
        if ntk_aware_scaling:
            base = RoPE.ntk_aware_base_scaling(base, head_dim, freq_cfg["ctx_len"], freq_cfg["og_ctx_len"])
        theta = 1 / base ** (2 * (torch.arange(0, head_dim // 2, dtype=dtype)) / head_dim)
        turns = freq_cfg["og_ctx_len"] / (2 * torch.pi / theta)  # full cycles inside the original context
        lo, hi, f = freq_cfg["alpha"], freq_cfg["beta"], freq_cfg["factor"]
        slow = theta / f
        blend = ((turns - lo) / (hi - lo)).clamp(0, 1)
        mixed = (1 - blend) * slow + blend * theta
        outside = torch.where(turns < lo, slow, theta)
        return torch.where((turns >= lo) & (turns <= hi), mixed, outside)

    @staticmethod
    def compute_angles(base, head_dim, ctx_len, smooth_scaling_cfg=None, ntk_aware_scaling=True, rotation_factor=1.0, dtype=torch.float32):
        """cos/sin tables, fp32, shape (ctx_len, rotated width), half-split layout [a | a] (reference rope.py:97-168)."""
        assert head_dim % 2 == 0, "head dim must be divisible by 2 as we have d/2 pairs of angles θi"
        assert dtype == torch.float32, "for now enforcing dtype as float32 as arg rather than .float() again"
        if rotation_factor != 1.0:
            head_dim = RoPE.partial_rotation(head_dim, rotation_factor)
        if smooth_scaling_cfg is not None:
            theta = RoPE.wavelength_scaling(base, head_dim, smooth_scaling_cfg, ntk_aware_scaling, dtype)
        else:
            theta = 1.0 / base ** (2 * (torch.arange(0, head_dim // 2, dtype=dtype)) / head_dim)
        ang = torch.outer(torch.arange(0, ctx_len, dtype=dtype), theta)
        ang = torch.cat([ang, ang], dim=-1)
        return torch.cos(ang), torch.sin(ang)

    @staticmethod
    def rotate_half(x):
        return _half_turn(x)

    @staticmethod
    def apply(x, cos, sin, position_ids=None):
        """RoPE on (b, heads, s, head_dim) with tables (ctx_len, R); R < head_dim = partial rotation; ``position_ids`` (b, s)
        picks table rows per token, otherwise rows 0..s-1 (reference rope.py:180-243)."""
        b, n_head, seq_length, head_dim = x.shape
        assert head_dim % 2 == 0, "head dim must be divisible by 2 as we need pairs"
        if x.is_cuda:
            cos_t, sin_t = _device_tables(cos, sin, x.device)
            idx = None
            if position_ids is not None:
                if tuple(position_ids.shape) != (b, seq_length):
                    raise ValueError(f"position_ids must be (b, s) = {(b, seq_length)}, got {tuple(position_ids.shape)}")
                idx = position_ids.to(device=x.device, dtype=torch.int32).reshape(-1).contiguous()
            return _RopeFn.apply(x if x.stride(-1) == 1 else x.contiguous(), cos_t, sin_t, idx)
        if position_ids is not None:
            c, s = cos[position_ids].unsqueeze(1).to(x.dtype), sin[position_ids].unsqueeze(1).to(x.dtype)
        else:
            c, s = cos[:seq_length, :].to(x.dtype), sin[:seq_length, :].to(x.dtype)
        return _host_rotate(x, c, s)

    @staticmethod
    def interleave_mrope_coeffs(cos, sin, mrope_section):
        """(3, b, s, half) coefficient rows per axis -> (b, s, half) with slots 1, 4, 7.. < 3*sec_h taken from H and
        2, 5, 8.. < 3*sec_w from W, the rest from T (MRoPE-I; reference rope.py:246-294)."""
        out_c, out_s = cos[0].clone(), sin[0].clone()
        for axis in (1, 2):
            slots = slice(axis, mrope_section[axis] * 3, 3)
            out_c[..., slots] = cos[axis, ..., slots]
            out_s[..., slots] = sin[axis, ..., slots]
        return out_c, out_s

    @staticmethod
    def apply_mrope(x, cos, sin, position_ids, mrope_section):
        """Multimodal RoPE on (b, heads, s, head_dim): ``position_ids`` (3, b, s) = (T, H, W) positions per token
        (reference rope.py:297-358)."""
        b, n_head, seq_length, head_dim = x.shape
        rotation_dim = cos.shape[-1]
        half = rotation_dim // 2
        if x.is_cuda:
            from llm_quest_amd import kernels_q35 as K35

            cos_t, sin_t = _device_tables(cos, sin, x.device)
            if tuple(position_ids.shape) != (3, b, seq_length):
                raise ValueError(f"position_ids must be (3, b, s) = {(3, b, seq_length)}, got {tuple(position_ids.shape)}")
            tc, ts = K35.mrope_table(cos_t, sin_t, position_ids.to(x.device), mrope_section)  # [b*s, R], exact gather
            idx = torch.arange(b * seq_length, dtype=torch.int32, device=x.device)
            return _RopeFn.apply(x if x.stride(-1) == 1 else x.contiguous(), tc, ts, idx)
        per_axis_c, per_axis_s = cos[:, :half][position_ids], sin[:, :half][position_ids]  # (3, b, s, half)
        c, s = RoPE.interleave_mrope_coeffs(per_axis_c, per_axis_s, mrope_section)
        c = torch.cat([c, c], dim=-1).unsqueeze(1).to(x.dtype)
        s = torch.cat([s, s], dim=-1).unsqueeze(1).to(x.dtype)
        return _host_rotate(x, c, s)


class VisionRoPE:
    """Axial 2-D RoPE for fixed-size images: half of the head rotates with the patch row, half with the patch column
    (reference rope.py:361-500)."""

    @staticmethod
    def compute_angles_2d(base, head_dim, height_patches, width_patches, num_frames=1, dtype=torch.float32):
        """(num_frames * H * W, head_dim) tables: cat([row * theta, col * theta]) duplicated, the spatial layout repeated per
        frame (reference rope.py:400-482)."""
        assert head_dim % 4 == 0, "head_dim must be divisible by 4 for 2D RoPE"
        half = head_dim // 2
        theta = 1.0 / (base ** (2 * torch.arange(0, half // 2, dtype=dtype) / half))
        rows = torch.arange(height_patches, dtype=dtype).repeat_interleave(width_patches)
        cols = torch.arange(width_patches, dtype=dtype).repeat(height_patches)
        ang = torch.cat([torch.outer(rows, theta), torch.outer(cols, theta)], dim=-1)
        if num_frames > 1:
            ang = ang.repeat(num_frames, 1)
        ang = torch.cat([ang, ang], dim=-1)
        return torch.cos(ang), torch.sin(ang)

    @staticmethod
    def apply(x, cos, sin, position_ids=None):
        return RoPE.apply(x, cos, sin, position_ids)
