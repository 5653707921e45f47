"""ViT encoder block pieces on HIP kernels -- API of ``llm_quest/multimodal/vision_transformer/vit_transformer_block.py``."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd.multimodal.vision_transformer.vit_attention import ViTMultiHeadAttention, bf16_cached, refuse_training

BF16, F32 = torch.bfloat16, torch.float32


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
        L.require_gpu(x)
        refuse_training(self, "LayerNorm")
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        if x2.dtype != F32:
            x2 = K.cast(x2, F32)
        return self.normalize(x2, F32).view(shp)


class GELU(nn.Module):
    """x * 0.5 * (1 + erf(x / sqrt(2))) (reference: :34-44)."""

    def __init__(self):
        super().__init__()

    def forward(self, x):
        L.require_gpu(x)
        if torch.is_grad_enabled() and x.requires_grad:
            raise NotImplementedError("stand-alone GELU backward is not wired; use ViTAdapter or eval mode")
        xb = x.contiguous() if x.dtype == BF16 else K.cast(x.contiguous(), BF16)
        y = K.gelu_fwd(xb)
        return y if x.dtype == BF16 else K.cast(y, x.dtype)


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
        L.require_gpu(x)
        refuse_training(self, "ViT FFN")
        shp = x.shape
        h = x.reshape(-1, shp[-1]).contiguous()
        h = h if h.dtype == BF16 else K.cast(h, BF16)
        f = self.hidden(h)
        w2 = bf16_cached(self, "w2", [self.layers[2].weight])
        return K.gemm(L.GEMM_NT, f, w2, bias=self.layers[2].bias.detach(), out_dtype=x.dtype).view(shp)


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
            x2 = K.gemm(L.GEMM_NT, ctx, wo, bias=self.att.out_proj.bias.detach(), residual=x2d, out_dtype=F32)
        h = self.ln_2.normalize(x2, BF16)
        f = self.ffn.hidden(h)
        w2 = bf16_cached(self.ffn, "w2", [self.ffn.layers[2].weight])
        if p > 0:
            return K.dropout(K.gemm(L.GEMM_NT, f, w2, bias=self.ffn.layers[2].bias.detach(), out_dtype=F32), p, *rng.draw(), residual=x2)
        return K.gemm(L.GEMM_NT, f, w2, bias=self.ffn.layers[2].bias.detach(), residual=x2, out_dtype=F32)

    def forward(self, x):
        L.require_gpu(x)
        refuse_training(self, "ViTTransformerBlock")
        B, S, d = x.shape
        x2 = x.reshape(B * S, d).contiguous()
        x2 = x2 if x2.dtype == F32 else K.cast(x2, F32)
        return self.run(x2, B, S).view(B, S, d)
