"""Process-global cache of the causal mask and RoPE tables (API of ``llm_quest/common/buffers.py``).

The reference keys its cache by the table's parameters only; the tables are then whatever device the FIRST model of the process was built on, and a second model
built under ``with torch.device("cuda")`` inherits CPU tables that no ``.to()`` ever moves.  Here the device the tables would be created on (the default device in
force at the call) is part of the key.

The HIP attention kernels never read the (ctx, ctx) mask -- causality is computed from indices -- but the buffer is
still produced so ``state_dict()`` / attribute access match the reference (long-context row).
"""

import torch

from .rope import RoPE


class GlobalBuffers:
    _mask_buffer = {}
    _rope_buffer = {}

    @staticmethod
    def get_causal_mask(ctx_len):
        """bool (ctx, ctx), True = masked (strict upper triangle); cached per ctx_len (buffers.py:25-37)."""
        key = (ctx_len, str(torch.get_default_device()))
        m = GlobalBuffers._mask_buffer.get(key)
        if m is None:
            m = torch.ones(ctx_len, ctx_len, dtype=torch.bool).triu_(1)
            GlobalBuffers._mask_buffer[key] = m
        return m

    @staticmethod
    def get_rope_params(ctx_len, rope_base, head_dim, smooth_scaling_cfg=None, rotation_factor=1.0):
        key = (ctx_len, rope_base, head_dim, smooth_scaling_cfg, rotation_factor, str(torch.get_default_device()))
        if key not in GlobalBuffers._rope_buffer:
            GlobalBuffers._rope_buffer[key] = RoPE.compute_angles(
                base=rope_base, head_dim=head_dim, ctx_len=ctx_len, smooth_scaling_cfg=smooth_scaling_cfg, rotation_factor=rotation_factor
            )
        return GlobalBuffers._rope_buffer[key]
