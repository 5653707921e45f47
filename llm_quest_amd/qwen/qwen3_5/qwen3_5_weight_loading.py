"""Hugging Face -> this package's parameter names for Qwen3.5, text stack and vision tower (API of
``llm_quest/qwen/qwen3_5/qwen3_5_weight_loading.py:22-178``).

As for Qwen3 (``qwen3_weight_loading.py`` next door) the hub download is replaced by a local source: a dict of tensors, a
``.safetensors`` file or a directory of shards.  ``load_state_dict`` copies into the existing parameter storage, so the bf16 block
arenas the fused projections read (``arena.py``) and the fp32 stragglers (``log_A``, ``post_norm.weight``) are filled in place and
nothing has to be re-packed afterwards; checkpoint tensors are cast to the dtype each parameter already has.
"""

from collections import namedtuple

import torch

from llm_quest_amd.utils import convert_weights, handle_weight_tying, report_loading_status, resolve_checkpoint

_TEXT_ROOT = "model.language_model."
_VISION_ROOT = "model.visual."
_MTP_ROOT = "mtp."  # multi-token-prediction head: not part of either model (upstream ignores it too)

# per-block tensors: HF suffix -> ours.  HF's q_proj already holds [query | gate] rows per head, which is the layout
# ``w_queries_gate`` has here, so every entry is a rename and none is a re-pack.
_TEXT_BLOCK = {
    "input_layernorm.weight": "norm1.scale",
    "post_attention_layernorm.weight": "norm2.scale",
    # GatedAttention layers (every ``linear_sdpa_ratio``-th block)
    "self_attn.q_proj.weight": "att.w_queries_gate.weight",
    "self_attn.k_proj.weight": "att.w_keys.weight",
    "self_attn.v_proj.weight": "att.w_values.weight",
    "self_attn.o_proj.weight": "att.out_proj.weight",
    "self_attn.q_norm.weight": "att.q_norm.scale",
    "self_attn.k_norm.weight": "att.k_norm.scale",
    # FusedGatedDeltaNet layers
    "linear_attn.A_log": "att.log_A",
    "linear_attn.dt_bias": "att.dt_bias",
    "linear_attn.in_proj_qkv.weight": "att.w_qkv.weight",
    "linear_attn.in_proj_z.weight": "att.w_gate.weight",
    "linear_attn.in_proj_b.weight": "att.w_beta.weight",
    "linear_attn.in_proj_a.weight": "att.w_alpha.weight",
    "linear_attn.conv1d.weight": "att.conv1d.weight",
    "linear_attn.norm.weight": "att.post_norm.weight",
    "linear_attn.out_proj.weight": "att.out_proj.weight",
    # SwiGLU MLP
    "mlp.gate_proj.weight": "ffn.lin_gate.weight",
    "mlp.up_proj.weight": "ffn.lin1.weight",
    "mlp.down_proj.weight": "ffn.lin2.weight",
}

_VISION_BLOCK = {  # weight and bias alike (rules end at the dot); norm1 / norm2 carry the same names on both sides
    "attn.qkv.": "att.qkv.",
    "attn.proj.": "att.proj.",
    "mlp.linear_fc1.": "ffn.lin1.",
    "mlp.linear_fc2.": "ffn.lin2.",
}


def get_remapping_rules():
    """(HF substring, our substring) pairs for the text stack, applied in order by ``convert_weights``
    (reference qwen3_5_weight_loading.py:22-58)."""
    rules = [
        (_TEXT_ROOT + "embed_tokens.weight", "emb_dict.weight"),
        (_TEXT_ROOT + "norm.weight", "final_norm.scale"),
        (_TEXT_ROOT + "layers.", "trf_blocks."),
    ]
    rules += [("." + hf, "." + ours) for hf, ours in _TEXT_BLOCK.items()]
    return rules


def get_vision_remapping_rules():
    """Same for the vision tower and its merge adapter (reference qwen3_5_weight_loading.py:61-82)."""
    rules = [
        (_VISION_ROOT + "patch_embed.proj.", "patch_embed.conv_proj."),
        (_VISION_ROOT + "pos_embed.", "pos_embed."),
        (_VISION_ROOT + "blocks.", "blocks."),
    ]
    rules += [("." + hf, "." + ours) for hf, ours in _VISION_BLOCK.items()]
    rules += [(_VISION_ROOT + "merger." + hf, "merge_adapter." + ours)
              for hf, ours in (("norm.", "norm."), ("linear_fc1.", "lin1."), ("linear_fc2.", "lin2."))]
    return rules


def load_qwen3_5_text_weights(model, model_cfg, source=None, verbose=True):
    """Text-only model from a full Qwen3.5 checkpoint: ``model.visual.*`` and ``mtp.*`` are skipped
    (reference qwen3_5_weight_loading.py:85-117)."""
    hf_state_dict = resolve_checkpoint(source, model_cfg)
    converted = convert_weights(hf_state_dict, model.state_dict(), get_remapping_rules(),
                                ignored_prefixes=(_VISION_ROOT, _MTP_ROOT))
    with torch.no_grad():
        result = model.load_state_dict(converted, strict=False)
        handle_weight_tying(model)
    if verbose:
        report_loading_status(model, result, converted)
    return model


_LoadResult = namedtuple("_LoadResult", ["missing_keys", "unexpected_keys"])


def load_qwen3_5_vlm_weights(model, model_cfg, source=None, verbose=True):
    """Text stack + vision tower of a ``Qwen3_5VLM`` from one checkpoint; the two halves are converted separately and reported
    together, prefixed with the attribute they live under (reference qwen3_5_weight_loading.py:120-178)."""
    hf_state_dict = resolve_checkpoint(source, model_cfg)
    halves = (
        ("language_model", model.language_model, get_remapping_rules(), (_VISION_ROOT, _MTP_ROOT)),
        ("vision_model", model.vision_model, get_vision_remapping_rules(), (_TEXT_ROOT, _MTP_ROOT)),
    )
    missing, unexpected, loaded = [], [], {}
    with torch.no_grad():
        for attr, sub, rules, ignored in halves:
            converted = convert_weights(hf_state_dict, sub.state_dict(), rules, ignored_prefixes=ignored)
            result = sub.load_state_dict(converted, strict=False)
            missing += [f"{attr}.{k}" for k in result.missing_keys]
            unexpected += [f"{attr}.{k}" for k in result.unexpected_keys]
            loaded.update({f"{attr}.{k}": v for k, v in converted.items()})
        handle_weight_tying(model.language_model)
    if verbose:
        report_loading_status(model, _LoadResult(missing, unexpected), loaded)
    return model
