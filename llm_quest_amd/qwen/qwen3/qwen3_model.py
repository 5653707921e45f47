"""Dense Qwen3 model on HIP kernels -- API of ``llm_quest/qwen/qwen3/qwen3_model.py`` (Qwen3Model).

Extensions over the reference signature (both needed by BASELINE config 4):
  * ``forward(..., input_embedded=False)``: ``x`` may already be embeddings (b, s, emb) -- the early-fusion entry that
    upstream only has on GPTModel (gpt_model.py:43-51);
  * ``forward_hidden`` / ``lm_loss``: the engine's fast path -- final-normed hidden states, then the tied LM head and
    cross entropy on just the rows that feed the loss (no (b, s, vocab) logits for positions nobody reads).
"""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import ops
from llm_quest_amd.arena import ParamArena
from llm_quest_amd.common.buffers import GlobalBuffers
from llm_quest_amd.qwen.qwen3.qwen3_attention import PytorchRMSNorm
from llm_quest_amd.qwen.qwen3.qwen3_transformer_block import TransformerBlock


class _Embedding(nn.Embedding):
    def forward(self, ids):
        L.require_gpu(ids)
        return ops.EmbeddingFn.apply(ids, self, self.weight)


class _OutHead(nn.Module):
    """Bias-free projection to the vocabulary; ``weight`` may be the embedding matrix itself (tied)."""

    def __init__(self, weight):
        super().__init__()
        self.weight = weight

    def forward(self, x):
        L.require_gpu(x)
        return ops.LinearFn.apply(x, self, self.weight, None, False)


class Qwen3Model(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.tie_embeddings = cfg["tie_embeddings"]
        self.gradient_checkpointing = cfg.get("gradient_checkpointing", False)
        self.emb_dict = _Embedding(cfg["vocab_size"], cfg["emb_dim"], dtype=cfg["dtype"])
        self.trf_blocks = nn.ModuleList([TransformerBlock(cfg, i) for i in range(cfg["n_layers"])])
        self.final_norm = PytorchRMSNorm(cfg["emb_dim"], dtype=cfg["dtype"])
        if self.tie_embeddings:
            # shared matrix re-initialised Xavier-uniform, as upstream does for pre-training (qwen3_model.py:36-45)
            self.out_head = _OutHead(self.emb_dict.weight)
            nn.init.xavier_uniform_(self.out_head.weight)
        else:
            w = nn.Parameter(torch.empty(cfg["vocab_size"], cfg["emb_dim"], dtype=cfg["dtype"]))
            nn.init.kaiming_uniform_(w, a=5**0.5)
            self.out_head = _OutHead(w)
        cos, sin = GlobalBuffers.get_rope_params(cfg["context_length"], cfg["rope_base"], cfg["head_dim"])
        self.register_buffer("mask", GlobalBuffers.get_causal_mask(cfg["context_length"]))
        self.register_buffer("cos", cos)
        self.register_buffer("sin", sin)
        self._arenas_built = False

    # ------------------------------------------------------------------ arenas: one per block + one for the rest
    def _build_arenas(self):
        if self._arenas_built:
            return
        for blk in self.trf_blocks:
            ar = ParamArena(list(blk.named_parameters()))
            for m in blk.modules():
                object.__setattr__(m, "_arena", ar)
        top = [("emb_dict.weight", self.emb_dict.weight), ("final_norm.weight", self.final_norm.weight)]
        if not self.tie_embeddings:
            top.append(("out_head.weight", self.out_head.weight))
        ar = ParamArena(top)
        for m in (self.emb_dict, self.final_norm, self.out_head):
            object.__setattr__(m, "_arena", ar)
        object.__setattr__(self, "_top_arena", ar)
        self._arenas_built = True

    def arenas(self):
        """Gradient buckets in backward-completion order: top (LM head) first is NOT complete until the embedding
        backward, so the order is blocks last->first, then the top arena."""
        self._build_arenas()
        return [blk._arena for blk in reversed(self.trf_blocks)] + [self._top_arena]

    # ------------------------------------------------------------------ forward paths
    def forward_hidden(self, x, attn_mask=None, position_ids=None, input_embedded=False, keep_rows=None):
        """Blocks + final norm: (b, s) ids or (b, s, emb) embeddings -> (b, s, emb).  ``keep_rows`` = (lo, hi): the caller reads only rows
        [lo, hi) of every sample (the early-fusion loss: the 512 rows that predict text tokens) -- the LAST block then runs its FFN half on those
        rows only and (b, hi - lo, emb) comes back; the kept rows are the same bits (ops.block_forward)."""
        self._build_arenas()
        L.require_gpu(x)
        if not input_embedded:
            x = self.emb_dict(x)
        elif x.dtype != self.emb_dict.weight.dtype:
            raise TypeError(f"embedded input must be {self.emb_dict.weight.dtype}, got {x.dtype}")
        B, S, _ = x.shape
        rt = ops.make_runtime(B, S, x.device, self.cos, self.sin, attn_mask, position_ids)
        use_ckpt = self.gradient_checkpointing and self.training and torch.is_grad_enabled()
        if keep_rows is not None and tuple(keep_rows) == (0, S):
            keep_rows = None
        last = len(self.trf_blocks) - 1
        for i, blk in enumerate(self.trf_blocks):
            if keep_rows is not None and i == last:
                x = ops.run_block(blk, x, rt, recompute=use_ckpt, rows=(int(keep_rows[0]), int(keep_rows[1])))
            elif use_ckpt:  # activation recomputation inside the block's own autograd node (ops.Qwen3BlockFn)
                x = ops.run_block(blk, x, rt, recompute=True)
            else:
                x = blk(x, self.mask, self.cos, self.sin, attn_mask, None, position_ids, _runtime=rt)
        return self.final_norm(x)

    def forward(self, x, attn_mask=None, kv_cache=None, position_ids=None, input_embedded=False):
        """Logits (b, s, vocab) in the model dtype (reference: qwen3_model.py:60-94)."""
        if kv_cache is not None:  # inference: prefill / one-token decode through utils.KVCache
            from llm_quest_amd import ops_decode

            return ops_decode.qwen3_forward_cached(self, x, kv_cache, attn_mask, position_ids, input_embedded)
        return self.out_head(self.forward_hidden(x, attn_mask, position_ids, input_embedded))

    def lm_loss(self, hidden_rows, targets):
        """Mean CE (ignore_index=-100) of the tied head on ``hidden_rows`` (rows, emb) vs ``targets`` (rows,)."""
        self._build_arenas()
        h = hidden_rows if hidden_rows.is_contiguous() else hidden_rows.contiguous()
        t = targets.reshape(-1).contiguous()
        return ops.LMHeadLossFn.apply(h, t, self.out_head, self.out_head.weight, torch.is_grad_enabled())
