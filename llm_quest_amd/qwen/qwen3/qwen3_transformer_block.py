"""Qwen3 SwiGLU FFN and transformer block -- API of ``llm_quest/qwen/qwen3/qwen3_transformer_block.py``."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd import ops
from llm_quest_amd.qwen.qwen3.qwen3_attention import GroupedQueryAttention, PytorchRMSNorm


class _FFNFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, ffn, keep, *params):
        arena = ops.arena_for(ffn)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        F_ = ffn.lin1.weight.shape[0]
        gu = K.gemm(L.GEMM_NT, x2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
        a = K.swiglu_fwd(gu, F_)
        y = K.gemm(L.GEMM_NT, a, ffn.lin2.weight)
        ctx.ffn, ctx.saved, ctx.shp = ffn, (x2, gu, a) if keep else None, shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        ffn = ctx.ffn
        arena = ops.arena_for(ffn)
        x2, gu, a = ctx.saved
        F_ = ffn.lin1.weight.shape[0]
        dy2 = dy.reshape(x2.shape[0], -1).contiguous()
        da = K.dgrad(dy2, ffn.lin2.weight)
        ops._wgrad(arena, ffn.lin2.weight, None, dy2, a)
        dgu = K.swiglu_bwd(gu, da, F_)
        dx = K.dgrad(dgu, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
        ops._wgrad(arena, ffn.lin1.weight, ffn.lin_gate.weight, dgu, x2)
        ctx.saved = None
        return (dx.view(ctx.shp), None, None) + (None,) * len(ffn._param_list)


class FFN(nn.Module):
    """lin2(lin1(x) * silu(lin_gate(x))), no biases (reference: qwen3_transformer_block.py:7-53)."""

    def __init__(self, cfg):
        super().__init__()
        # lin1 then lin_gate: adjacent in the arena -> one [2*hidden, emb] GEMM
        self.lin1 = nn.Linear(cfg["emb_dim"], cfg["hidden_dim"], dtype=cfg["dtype"], bias=False)
        self.lin_gate = nn.Linear(cfg["emb_dim"], cfg["hidden_dim"], dtype=cfg["dtype"], bias=False)
        self.lin2 = nn.Linear(cfg["hidden_dim"], cfg["emb_dim"], dtype=cfg["dtype"], bias=False)

    def forward(self, x):
        L.require_gpu(x)
        if not hasattr(self, "_param_list"):
            object.__setattr__(self, "_param_list", list(self.parameters()))
        return _FFNFn.apply(x, self, torch.is_grad_enabled(), *self._param_list)


class TransformerBlock(nn.Module):
    """Pre-norm block: x + att(norm1(x)); x + ffn(norm2(x)) (reference: qwen3_transformer_block.py:56-103).

    Runs as ONE autograd node (ops.Qwen3BlockFn): 10 kernels forward, ~20 backward, residual adds fused into the GEMM
    epilogues and into the RMSNorm backward.
    """

    def __init__(self, cfg, layer_idx):
        super().__init__()
        self.att = GroupedQueryAttention(
            d_in=cfg["emb_dim"], num_heads=cfg["n_heads"], num_kv_groups=cfg["num_kv_groups"], head_dim=cfg["head_dim"],
            dtype=cfg["dtype"], layer_idx=layer_idx,
        )
        self.norm1 = PytorchRMSNorm(cfg["emb_dim"], dtype=cfg["dtype"])
        self.norm2 = PytorchRMSNorm(cfg["emb_dim"], dtype=cfg["dtype"])
        self.ffn = FFN(cfg)

    def forward(self, x, mask, cos, sin, attn_mask=None, kv_cache=None, position_ids=None, _runtime=None):
        B, S, _ = x.shape
        if kv_cache is not None:  # inference (reference qwen3_transformer_block.py:99-100 -> qwen3_attention.py:117-118), no autograd
            from llm_quest_amd import ops_decode as OD

            L.require_gpu(x)
            h = x.reshape(B * S, -1)
            h = (h if h.dtype == torch.bfloat16 else K.cast(h.contiguous(), torch.bfloat16)).contiguous()
            pos = OD.cached_positions(kv_cache, B, S, x.device, position_ids, reference_default=True)
            km = OD.cached_key_mask(attn_mask, kv_cache, B, S, x.device)
            y = OD.block_cached(self, h, B, S, cos, sin, pos, km, kv_cache)
            return (y if y.dtype == x.dtype else K.cast(y, x.dtype)).view(B, S, -1)
        rt = _runtime if _runtime is not None else ops.make_runtime(B, S, x.device, cos, sin, attn_mask, position_ids)
        return ops.run_block(self, x, rt)
