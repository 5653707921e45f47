"""GPT-2 plumbing model for BASELINE config 1 (CPU forward, batch 4 x seq 128) -- API of ``llm_quest/gpt/gpt_model.py``.

Config 1 is explicitly a host-side smoke test of the boundary ("plumbing, no GPU"), so this model is plain PyTorch and
is NOT part of the HIP hot path; it exists so the engine / VLM loops can be exercised end to end without a GPU (the
reference's own VLM loop is written against this model: ``emb_dict``, ``pos_emb_dict``, ``input_embedded``).
State-dict keys follow the reference (``trf_blocks.N.att.w_queries`` ..., ``ln_1.scale`` ..., ``ffn.layers.0`` ..., ``out``).
"""

import math

import torch
import torch.nn as nn


class LayerNorm(nn.Module):
    def __init__(self, emb_dim):
        super().__init__()
        self.eps = 1e-5
        self.scale = nn.Parameter(torch.ones(emb_dim))
        self.shift = nn.Parameter(torch.zeros(emb_dim))

    def forward(self, x):
        mu = x.mean(dim=-1, keepdim=True)
        sd = torch.std(x, dim=-1, keepdim=True, unbiased=False)
        return self.scale * ((x - mu) / (sd + self.eps)) + self.shift  # eps on sigma (gpt_transformer_block.py:35-39)


class GELU(nn.Module):
    def forward(self, x):
        return x * 0.5 * (1 + torch.erf(x / math.sqrt(2)))


class FFN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.layers = nn.Sequential(nn.Linear(cfg["emb_dim"], 4 * cfg["emb_dim"]), GELU(), nn.Linear(4 * cfg["emb_dim"], cfg["emb_dim"]))

    def forward(self, x):
        return self.layers(x)


class MultiHeadAttention(nn.Module):
    def __init__(self, d_in, d_out, dropout, ctx_len, num_heads, qkv_bias=False, layer_idx=None):
        super().__init__()
        if d_out % num_heads != 0:
            raise ValueError("d_out must be divisible by num_heads")
        self.d_out, self.num_heads, self.head_dim = d_out, num_heads, d_out // num_heads
        self.att_scaling = self.head_dim**-0.5
        self.layer_idx = layer_idx
        self.w_queries = nn.Linear(d_in, d_out, bias=qkv_bias)
        self.w_keys = nn.Linear(d_in, d_out, bias=qkv_bias)
        self.w_values = nn.Linear(d_in, d_out, bias=qkv_bias)
        self.dropout = nn.Dropout(dropout)
        self.register_buffer("mask", torch.ones(ctx_len, ctx_len).triu_(1).bool())
        self.out_proj = nn.Linear(d_out, d_out)

    def forward(self, x, attn_mask=None, kv_cache=None):
        if kv_cache is not None:
            raise NotImplementedError("KV-cache decoding is out of scope")
        b, s, _ = x.shape
        split = lambda t: t.view(b, s, self.num_heads, self.head_dim).transpose(1, 2)
        q, k, v = split(self.w_queries(x)), split(self.w_keys(x)), split(self.w_values(x))
        scores = (q @ k.mT) * self.att_scaling
        blocked = self.mask[:s, :s]
        if attn_mask is not None:
            blocked = blocked.view(1, 1, s, s) | ~attn_mask.view(b, 1, 1, s)
        scores = scores.masked_fill(blocked, torch.finfo(scores.dtype).min / 2)  # finite fill (gpt_attention.py:199-200)
        ctx = self.dropout(torch.softmax(scores, dim=-1)) @ v
        return self.out_proj(ctx.transpose(1, 2).contiguous().view(b, s, self.d_out))


class TransformerBlock(nn.Module):
    def __init__(self, cfg, layer_idx=None):
        super().__init__()
        self.att = MultiHeadAttention(cfg["emb_dim"], cfg["emb_dim"], cfg["drop_rate"], cfg["context_length"], cfg["n_heads"], cfg["qkv_bias"], layer_idx)
        self.ln_1 = LayerNorm(cfg["emb_dim"])
        self.ln_2 = LayerNorm(cfg["emb_dim"])
        self.ffn = FFN(cfg)
        self.dropout = nn.Dropout(cfg["drop_rate"])

    def forward(self, x, attn_mask=None, kv_cache=None):
        x = x + self.dropout(self.att(self.ln_1(x), attn_mask, kv_cache))
        return x + self.dropout(self.ffn(self.ln_2(x)))


class GPTModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.emb_dict = nn.Embedding(cfg["vocab_size"], cfg["emb_dim"])
        self.pos_emb_dict = nn.Embedding(cfg["context_length"], cfg["emb_dim"])
        self.dropout = nn.Dropout(cfg["drop_rate"])
        self.trf_blocks = nn.ModuleList([TransformerBlock(cfg, layer_idx=i) for i in range(cfg["n_layers"])])
        self.final_ln = LayerNorm(cfg["emb_dim"])
        self.out = nn.Linear(cfg["emb_dim"], cfg["vocab_size"], bias=False)

    def forward(self, x, attn_mask=None, kv_cache=None, last_token_only=False, input_embedded=False, position_ids=None):
        if kv_cache is not None:
            raise NotImplementedError("KV-cache decoding is out of scope")
        b, s = x.shape[:2]
        if not input_embedded:
            x = self.emb_dict(x)
            if position_ids is None:
                position_ids = torch.arange(s, device=x.device).unsqueeze(0)
            x = x + self.pos_emb_dict(position_ids)
        x = self.dropout(x)
        for blk in self.trf_blocks:
            x = blk(x, attn_mask, kv_cache)
        x = self.final_ln(x)
        if last_token_only:
            assert attn_mask is not None, "attn_mask are needed for last_token_only=True"
            return self.out(x[torch.arange(b), attn_mask.sum(dim=-1) - 1, :])
        return self.out(x)
