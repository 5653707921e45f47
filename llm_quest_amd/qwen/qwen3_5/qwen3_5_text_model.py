"""Qwen3.5 hybrid text model on HIP kernels -- API of ``llm_quest/qwen/qwen3_5/qwen3_5_text_model.py`` (BASELINE config 5,
SURVEY.md section 8 row a24): ``FusedGatedDeltaNet``, ``MRoPEGatedAttention``, ``Qwen3_5TransformerBlock``, ``Qwen3_5TextModel``
with the reference's constructor keys, forward signatures and ``state_dict`` keys.

One autograd node per block (llm_quest_amd/ops_q35.py).  ``cache=Qwen3_5Cache`` decodes through ``ops_decode`` at every level
the reference takes it: the model (``qwen35_forward_cached``), a block, or a token mixer called on its own (``q35_block_cached`` / ``q35_mixer_cached``:
the same kernels, so the layers composed by hand equal the model bit for bit).  Extensions as on Qwen3Model: ``forward_hidden`` / ``lm_loss`` (LM head + cross entropy on
just the rows that feed the loss).
"""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import ops, ops_q35
from llm_quest_amd.arena import ParamArena
from llm_quest_amd.common.buffers import GlobalBuffers
from llm_quest_amd.qwen.qwen3.qwen3_attention import PytorchRMSNorm
from llm_quest_amd.qwen.qwen3.qwen3_model import _Embedding, _OutHead
from llm_quest_amd.qwen.qwen3.qwen3_transformer_block import FFN
from llm_quest_amd.qwen.qwen3_next.qwen3_next_attention import GatedAttention, ZeroCenteredRMSNorm


def _cached_step(x, cache, cos, sin, mrope_section, attn_mask, position_ids):
    """(rows [b*s, d], ops_decode.Q35Step) of a layer called with a ``Qwen3_5Cache`` (inference: no autograd graph is built)."""
    from llm_quest_amd import ops_decode

    L.require_gpu(x)
    b, s, d = x.shape
    st = ops_decode.Q35Step(cache, b, s, x.device, cos, sin, mrope_section, attn_mask, position_ids)
    return x.detach().reshape(b * s, d).contiguous(), st


class FusedGatedDeltaNet(nn.Module):
    """Gated delta net with fused QKV projection and one depthwise conv over it (reference :36-191)."""

    is_linear = True

    def __init__(self, cfg, layer_idx=None):
        super().__init__()
        self.layer_idx = layer_idx
        self.d_in = cfg["emb_dim"]
        self.num_qk_heads = cfg["linear_num_qk_heads"]
        self.num_v_heads = cfg["linear_num_value_heads"]
        self.qk_head_dim = cfg["linear_qk_head_dim"]
        self.vg_head_dim = cfg["linear_value_head_dim"]
        self.conv_kernel_size = cfg["linear_conv_kernel_size"]
        self.num_repeat = self.num_v_heads // self.num_qk_heads
        self.d_out = self.num_qk_heads * self.qk_head_dim
        self.d_out_vg = self.num_v_heads * self.vg_head_dim
        self.dtype = cfg["dtype"]
        fused = self.d_out * 2 + self.d_out_vg
        # w_qkv | w_gate | w_beta | w_alpha are adjacent in the block's arena: one projection GEMM
        self.w_qkv = nn.Linear(self.d_in, fused, bias=False, dtype=self.dtype)
        self.w_gate = nn.Linear(self.d_in, self.d_out_vg, bias=False, dtype=self.dtype)
        self.w_beta = nn.Linear(self.d_in, self.num_v_heads, bias=False, dtype=self.dtype)
        self.w_alpha = nn.Linear(self.d_in, self.num_v_heads, bias=False, dtype=self.dtype)
        A_init = torch.empty(self.num_v_heads, dtype=torch.float32).uniform_(0, 16)
        self.log_A = nn.Parameter(torch.log(A_init))  # fp32, as upstream
        self.dt_bias = nn.Parameter(torch.ones(self.num_v_heads, dtype=self.dtype))
        self.conv1d = nn.Conv1d(fused, fused, kernel_size=self.conv_kernel_size, bias=False, padding=self.conv_kernel_size - 1, groups=fused, dtype=self.dtype)
        self.activation = nn.SiLU()
        self.post_norm = PytorchRMSNorm(self.vg_head_dim, dtype=torch.float32)
        self.out_proj = nn.Linear(self.d_out_vg, self.d_in, bias=False, dtype=self.dtype)

    def forward(self, x, attn_mask=None, cache=None):
        """x (b, s, d_in); attn_mask (b, s), 1 = real token.  Unlike upstream, ``x`` is not modified in place (the mask is applied
        to a copy; inside the block the masked tensor is the block's own temporary either way)."""
        b, s, _ = x.shape
        if cache is not None:  # conv state + recurrent state of layer ``layer_idx`` (reference :104-106, 131-160)
            from llm_quest_amd import ops_decode

            rows, st = _cached_step(x, cache, None, None, None, attn_mask, None)
            mix = ops_decode.q35_mixer_cached(self, ops_q35.arena_for_bf16(self), rows, st, self.layer_idx)
            return st.lin(mix, self.out_proj.weight).view(b, s, -1)
        rt = ops_q35.Runtime(b, s, None if attn_mask is None else attn_mask.to(device=x.device, dtype=torch.uint8).contiguous(), None, None, None)
        return ops_q35.run_mixer(self, x, rt)


class MRoPEGatedAttention(GatedAttention):
    """GatedAttention with interleaved multimodal RoPE (reference :194-267)."""

    def __init__(self, cfg, layer_idx=None):
        super().__init__(cfg)
        self.layer_idx = layer_idx
        self.mrope_section = cfg["mrope_section"]

    def forward(self, x, mask, cos, sin, position_ids=None, attn_mask=None, cache=None):
        b, s, _ = x.shape
        if cache is not None:  # K / V rows of layer ``layer_idx`` (reference :236-238)
            from llm_quest_amd import ops_decode

            rows, st = _cached_step(x, cache, cos, sin, self.mrope_section, attn_mask, position_ids)
            mix = ops_decode.q35_mixer_cached(self, ops_q35.arena_for_bf16(self), rows, st, self.layer_idx)
            return st.lin(mix, self.out_proj.weight).view(b, s, -1)
        rt = ops_q35.make_runtime(b, s, x.device, cos, sin, attn_mask=attn_mask, position_ids=position_ids, mrope_section=self.mrope_section)
        return ops_q35.run_mixer(self, x, rt)


class Qwen3_5TransformerBlock(nn.Module):
    """Pre-norm block, FusedGatedDeltaNet on layers with (layer_idx + 1) % linear_sdpa_ratio != 0, MRoPEGatedAttention otherwise
    (reference :270-325).  Runs as ONE autograd node (ops_q35.Qwen35BlockFn)."""

    def __init__(self, cfg, layer_idx):
        super().__init__()
        interval = cfg["linear_sdpa_ratio"]
        self.is_linear = bool((layer_idx + 1) % interval)
        self.att = FusedGatedDeltaNet(cfg, layer_idx=layer_idx) if self.is_linear else MRoPEGatedAttention(cfg, layer_idx=layer_idx)
        self.norm1 = ZeroCenteredRMSNorm(cfg["emb_dim"], dtype=cfg["dtype"])
        self.norm2 = ZeroCenteredRMSNorm(cfg["emb_dim"], dtype=cfg["dtype"])
        self.ffn = FFN(cfg)
        self.mrope_section = cfg["mrope_section"]

    def forward(self, x, mask, cos, sin, position_ids=None, attn_mask=None, cache=None, _runtime=None):
        b, s, _ = x.shape
        if cache is not None:  # reference :296-325 with the hybrid cache
            from llm_quest_amd import ops_decode

            rows, st = _cached_step(x, cache, cos, sin, self.mrope_section, attn_mask, position_ids)
            return ops_decode.q35_block_cached(self, rows, st).view(b, s, -1)
        rt = _runtime
        if rt is None:
            rt = ops_q35.make_runtime(b, s, x.device, cos, sin, attn_mask=attn_mask, position_ids=position_ids, mrope_section=self.mrope_section)
        return ops_q35.run_block(self, x, rt)


class Qwen3_5TextModel(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        if cfg["dtype"] != torch.bfloat16:
            raise TypeError("the HIP path of Qwen3_5TextModel computes in bf16 (cfg['dtype'] of every Qwen3.5 config)")
        self.tie_embeddings = cfg["tie_embeddings"]
        self.mrope_section = cfg["mrope_section"]
        self.emb_dict = _Embedding(cfg["vocab_size"], cfg["emb_dim"], dtype=cfg["dtype"])
        self.trf_blocks = nn.ModuleList([Qwen3_5TransformerBlock(cfg, i) for i in range(cfg["n_layers"])])
        self.final_norm = ZeroCenteredRMSNorm(cfg["emb_dim"], dtype=cfg["dtype"])
        if self.tie_embeddings:
            self.out_head = _OutHead(self.emb_dict.weight)  # shared matrix, Xavier-uniform re-init as upstream (:366-372)
            nn.init.xavier_uniform_(self.out_head.weight)
        else:
            w = nn.Parameter(torch.empty(cfg["vocab_size"], cfg["emb_dim"], dtype=cfg["dtype"]))
            nn.init.kaiming_uniform_(w, a=5**0.5)
            self.out_head = _OutHead(w)
        mask = GlobalBuffers.get_causal_mask(cfg["context_length"])
        cos, sin = GlobalBuffers.get_rope_params(cfg["context_length"], cfg["rope_base"], cfg["head_dim"], rotation_factor=cfg["partial_rope_factor"])
        self.register_buffer("mask", ~mask)  # inverted (True = attend) as upstream; the kernels derive causality from indices
        self.register_buffer("cos", cos)
        self.register_buffer("sin", sin)
        self._arenas_built = False

    # ------------------------------------------------------------------ arenas: one (bf16) per block + one for the rest
    def _build_arenas(self):
        if self._arenas_built:
            return
        for blk in self.trf_blocks:
            ops_q35.arena_for_bf16(blk)
        top = [("emb_dict.weight", self.emb_dict.weight), ("final_norm.scale", self.final_norm.scale)]
        if not self.tie_embeddings:
            top.append(("out_head.weight", self.out_head.weight))
        ar = ParamArena(top)
        for m in (self.emb_dict, self.final_norm, self.out_head):
            object.__setattr__(m, "_arena", ar)
        object.__setattr__(self, "_top_arena", ar)
        self._arenas_built = True

    def arenas(self):
        """bf16 gradient buckets in backward-completion order (blocks last -> first, then embedding / head)."""
        self._build_arenas()
        return [blk._arena for blk in reversed(self.trf_blocks)] + [self._top_arena]

    def standalone_parameters(self):
        """The fp32 parameters that live outside the arenas (log_A, post_norm.weight of every GDN layer)."""
        return [p for p in self.parameters() if p.dtype != torch.bfloat16]

    # ------------------------------------------------------------------ forward paths
    def forward_hidden(self, x=None, attn_mask=None, inputs_embs=None, position_ids=None):
        self._build_arenas()
        if inputs_embs is not None:
            x = inputs_embs
            if x.dtype != self.emb_dict.weight.dtype:
                raise TypeError(f"inputs_embs must be {self.emb_dict.weight.dtype}, got {x.dtype}")
        else:
            x = self.emb_dict(x)
        L.require_gpu(x)
        B, S, _ = x.shape
        rt = ops_q35.make_runtime(B, S, x.device, self.cos, self.sin, attn_mask=attn_mask, position_ids=position_ids, mrope_section=self.mrope_section)
        for blk in self.trf_blocks:
            x = blk(x, self.mask, self.cos, self.sin, position_ids, attn_mask, None, _runtime=rt)
        return self.final_norm(x)

    def forward(self, x=None, attn_mask=None, inputs_embs=None, position_ids=None, cache=None):
        """Logits (b, s, vocab) bf16 (reference :388-417).  ``cache`` (utils.Qwen3_5Cache): inference, prefill / one-token steps."""
        if cache is not None:
            from llm_quest_amd import ops_decode

            return ops_decode.qwen35_forward_cached(self, x, cache, attn_mask, inputs_embs, position_ids)
        return self.out_head(self.forward_hidden(x, attn_mask, inputs_embs, position_ids))

    def lm_loss(self, hidden_rows, targets):
        """Mean CE (ignore_index=-100) of the (tied) head on ``hidden_rows`` (rows, emb) vs ``targets`` (rows,)."""
        self._build_arenas()
        h = hidden_rows if hidden_rows.is_contiguous() else hidden_rows.contiguous()
        return ops.LMHeadLossFn.apply(h, targets.reshape(-1).contiguous(), self.out_head, self.out_head.weight, torch.is_grad_enabled())
