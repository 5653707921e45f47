"""ViT encoder block pieces on HIP kernels -- API of ``llm_quest/multimodal/vision_transformer/vit_transformer_block.py``."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd.multimodal.vision_transformer.vit_attention import ViTMultiHeadAttention, bf16_cached, split3_cached

BF16, F32 = torch.bfloat16, torch.float32


import os as _os

# tile of the block's two d-wide projections (out_proj, FFN down: N = emb_dim).  At the frozen tower's 31 520 rows they are 372 tiles of 256 x 256 = 1.45 rounds of the
# chip; 0 = the library's choice (A/B knob, see profiles/r06_notes.md)
_TILE_SMALL_N = int(_os.environ.get("MI355_VIT_TILE_SMALLN", "0"))


class LayerNorm(nn.Module):
    """scale * (x - mean) / (population_std + eps) + shift, eps = 1e-5 added to sigma (reference: :12-31)."""

    def __init__(self, emb_dim):
        super().__init__()
        self.eps = 1e-5
        self.scale = nn.Parameter(torch.ones(emb_dim))
        self.shift = nn.Parameter(torch.zeros(emb_dim))

    def normalize(self, x2d_f32, out_dtype):
        return K.layernorm_fwd(x2d_f32, self.scale.detach(), self.shift.detach(), out_dtype=out_dtype, eps=self.eps)

    def forward(self, x):
        """fp32 statistics whatever the input dtype; output in x.dtype.  Autograd node over ``layernorm_fwd`` / ``layernorm_bwd``."""
        from llm_quest_amd.multimodal.vision_transformer import vit_train as T

        def fwd(t):
            x2 = T.as_f32_rows(t)
            y, mean, rsig = K.layernorm_fwd(x2, self.scale.detach(), self.shift.detach(), out_dtype=F32, eps=self.eps, want_stats=True)
            return T.like(y, t, t.shape[-1]), (x2, mean, rsig)

        def bwd(saved, dy):
            x2, mean, rsig = saved
            g = dy.reshape(x2.shape)
            g = g if g.dtype in (F32, BF16) else g.to(F32)
            return T.like(T._ln_bwd(self, x2, mean, rsig, g, None), dy, dy.shape[-1])

        return T.run_piece(self, x, fwd, bwd)


class GELU(nn.Module):
    """x * 0.5 * (1 + erf(x / sqrt(2))) (reference: :34-44)."""

    def __init__(self):
        super().__init__()

    def forward(self, x):
        from llm_quest_amd.multimodal.vision_transformer import vit_train as T

        def fwd(t):
            xb = t.contiguous() if t.dtype == BF16 else K.cast(t.contiguous(), BF16)
            pad = (-xb.numel()) % 8  # the element-wise kernels move 16-byte words
            if pad:
                xb = torch.cat([xb.reshape(-1), xb.new_zeros(pad)])
            y = K.gelu_fwd(xb)
            y = y[: t.numel()].view(t.shape) if pad else y
            return (y if t.dtype == BF16 else K.cast(y.contiguous(), t.dtype)), (xb, pad)

        def bwd(saved, dy):
            xb, pad = saved
            g = dy.contiguous() if dy.dtype == BF16 else K.cast(dy.contiguous(), BF16)
            if pad:
                g = torch.cat([g.reshape(-1), g.new_zeros(pad)])
            dx = K.gelu_bwd(xb, g.view(xb.shape))
            dx = dx[: dy.numel()].view(dy.shape) if pad else dx.view(dy.shape)
            return dx if dy.dtype == BF16 else K.cast(dx.contiguous(), dy.dtype)

        return T.run_piece(self, x, fwd, bwd)


class FFN(nn.Module):
    """Linear(d, 4d) -> GELU -> Linear(4d, d) with biases (reference: :47-67); key names ``layers.0`` / ``layers.2``."""

    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(cfg["emb_dim"], 4 * cfg["emb_dim"]), GELU(), nn.Linear(4 * cfg["emb_dim"], cfg["emb_dim"]))

    def hidden(self, h_bf16):
        """bf16 [M, d] -> gelu(h W1^T + b1) bf16 [M, 4d] (bias + GELU fused in the GEMM epilogue)."""
        w1 = bf16_cached(self, "w1", [self.layers[0].weight])
        return K.gemm(L.GEMM_NT, h_bf16, w1, bias=self.layers[0].bias.detach(), gelu=True)

    def forward(self, x):
        from llm_quest_amd.multimodal.vision_transformer import vit_train as T

        d = x.shape[-1]

        def fwd(t):
            y, saved = T.ffn_forward(self, T.as_bf16_rows(t), out_dtype=F32)
            return T.like(y, t, d), saved

        def bwd(saved, dy):
            return T.like(T.ffn_backward(self, saved, dy.reshape(-1, d)), dy, d)

        return T.run_piece(self, x, fwd, bwd)


class ViTTransformerBlock(nn.Module):
    """Pre-LN encoder block (reference: :70-127).  fp32 residual stream, bf16 MFMA operands; both residual adds are
    fused into the out_proj / FFN-down GEMM epilogues."""

    def __init__(self, cfg):
        super().__init__()
        self.att = ViTMultiHeadAttention(d_in=cfg["emb_dim"], d_out=cfg["emb_dim"], dropout=cfg["drop_rate"], num_heads=cfg["n_heads"], qkv_bias=cfg["qkv_bias"])
        self.ln_1 = LayerNorm(cfg["emb_dim"])
        self.ln_2 = LayerNorm(cfg["emb_dim"])
        self.ffn = FFN(cfg)
        self.dropout = nn.Dropout(cfg["drop_rate"])

    def run(self, x2d, B, S):
        """x2d fp32 [B*S, d] -> fp32 [B*S, d]."""
        p = self.dropout.p if self.training else 0.0
        h = self.ln_1.normalize(x2d, BF16)
        ctx = self.att.context(h, B, S)
        wo = bf16_cached(self.att, "wo", [self.att.out_proj.weight])
        if p > 0:  # x + dropout(.) (reference :117, :124): the residual add is fused into the dropout pass
            from llm_quest_amd import rng

            x2 = K.dropout(K.gemm(L.GEMM_NT, ctx, wo, bias=self.att.out_proj.bias.detach(), out_dtype=F32), p, *rng.draw(), residual=x2d)
        else:
            x2 = K.gemm(L.GEMM_NT, ctx, wo, bias=self.att.out_proj.bias.detach(), residual=x2d, out_dtype=F32, tile=_TILE_SMALL_N)
        h = self.ln_2.normalize(x2, BF16)
        f = self.ffn.hidden(h)
        w2 = bf16_cached(self.ffn, "w2", [self.ffn.layers[2].weight])
        if p > 0:
            return K.dropout(K.gemm(L.GEMM_NT, f, w2, bias=self.ffn.layers[2].bias.detach(), out_dtype=F32), p, *rng.draw(), residual=x2)
        return K.gemm(L.GEMM_NT, f, w2, bias=self.ffn.layers[2].bias.detach(), residual=x2, out_dtype=F32, tile=_TILE_SMALL_N)

    def run_f32(self, x2d, B, S):
        """x2d fp32 [B*S, d] -> fp32 [B*S, d] at the reference's fp32 precision (inference only: the frozen tower of the early-fusion step,
        multimodal/vlm_engine.py:99-104).  Every nn.Linear is ONE bf16 MFMA GEMM over split operands ([hi | lo | hi] x [hi | hi | lo], K' = 3K, fp32
        accumulation and output), attention runs on the exact-fp32 MFMA, LayerNorm / GELU / residual adds in fp32 as before."""
        if self.training and (self.dropout.p > 0 or self.att.dropout.p > 0):
            raise RuntimeError("ViTTransformerBlock.run_f32 is the frozen tower's inference path: call .eval() (dropout is active)")
        att, ffn = self.att, self.ffn
        # (round 6: the LayerNorms and the GELU projection write their fp32 results as split operands themselves -- K.split3 of an fp32 tensor reads 4 and writes
        # 6 bytes per element, 20 GB per step at the bench's batch; only the attention's context still takes the separate pass)
        ctx = att.context_f32(self.ln_1.normalize(x2d, "split3"), B, S)
        x2 = K.gemm(L.GEMM_NT, K.split3(ctx), split3_cached(att, "wo3", [att.out_proj.weight]), bias=att.out_proj.bias.detach(), residual=x2d, out_dtype=F32)
        h = self.ln_2.normalize(x2, "split3")
        f3 = K.gemm(L.GEMM_NT, h, split3_cached(ffn, "w1_3", [ffn.layers[0].weight]), bias=ffn.layers[0].bias.detach(), gelu=True, split3_out=True)
        return K.gemm(L.GEMM_NT, f3, split3_cached(ffn, "w2_3", [ffn.layers[2].weight]), bias=ffn.layers[2].bias.detach(), residual=x2, out_dtype=F32)

    def forward(self, x):
        """(b, s, d) -> (b, s, d) in x.dtype; trains stand-alone (one autograd node over ``vit_train.block_forward`` / ``block_backward``,
        dropout as configured), as the reference's module does (vit_transformer_block.py:106-127)."""
        from llm_quest_amd.multimodal.vision_transformer import vit_train as T

        B, S, d = x.shape

        def fwd(t):
            y, saved = T.block_forward(self, T.as_f32_rows(t), B, S, self.training)
            return T.like(y, t, d), saved

        def bwd(saved, dy):
            return T.like(T.block_backward(self, saved, T.as_f32_rows(dy), B, S), dy, d)

        return T.run_piece(self, x, fwd, bwd)
