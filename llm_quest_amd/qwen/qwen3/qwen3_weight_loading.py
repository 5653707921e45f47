"""Hugging Face -> this package's parameter names for Qwen3 (API of
``llm_quest/qwen/qwen3/qwen3_weight_loading.py``; the shared helpers live in ``llm_quest_amd/utils.py`` as upstream's do in ``utils.py``).

There is no network on the build / GPU boxes, so the download step is replaced by a local source: a dict of tensors or a
``.safetensors`` file / directory of shards that the caller already has.  Name mapping, shape checks, weight tying and the
loading report behave as upstream; the loaded parameters land in the model's arenas (``load_state_dict`` copies into the
existing views, so fused QKV / gate-up GEMMs see them without any re-packing).
"""

import torch

from llm_quest_amd.utils import convert_weights, handle_weight_tying, report_loading_status, resolve_checkpoint


def get_remapping_rules(model_cfg):
    """(HF substring, our substring) pairs, applied in order (reference qwen3_weight_loading.py:19-66)."""
    rules = [
        ("model.embed_tokens.weight", "emb_dict.weight"),
        ("model.norm.weight", "final_norm.weight"),
        ("model.layers.", "trf_blocks."),
        (".self_attn.q_proj.weight", ".att.w_queries.weight"),
        (".self_attn.k_proj.weight", ".att.w_keys.weight"),
        (".self_attn.v_proj.weight", ".att.w_values.weight"),
        (".self_attn.o_proj.weight", ".att.out_proj.weight"),
        (".self_attn.q_norm.weight", ".att.q_norm.weight"),
        (".self_attn.k_norm.weight", ".att.k_norm.weight"),
        (".input_layernorm.weight", ".norm1.weight"),
        (".post_attention_layernorm.weight", ".norm2.weight"),
    ]
    if not model_cfg["tie_embeddings"]:
        rules.append(("lm_head.weight", "out_head.weight"))
    if model_cfg.get("model_type", "dense") == "moe":
        raise NotImplementedError("MoE checkpoints are outside this package's scope (DESIGN.md section 7)")
    rules += [
        (".mlp.gate_proj.weight", ".ffn.lin_gate.weight"),
        (".mlp.up_proj.weight", ".ffn.lin1.weight"),
        (".mlp.down_proj.weight", ".ffn.lin2.weight"),
    ]
    return rules


def load_qwen3_weights(model, model_cfg, source=None, verbose=True):
    """Convert + load a Hugging Face Qwen3 checkpoint (reference qwen3_weight_loading.py:69-103).  ``source`` (a dict, a
    ``.safetensors`` file or a shard directory) overrides ``model_cfg["model_path"]``, which must itself be a local snapshot:
    this package never opens a network connection."""
    hf_state_dict = resolve_checkpoint(source, model_cfg)
    converted = convert_weights(hf_state_dict, model.state_dict(), get_remapping_rules(model_cfg))
    with torch.no_grad():
        result = model.load_state_dict(converted, strict=False)
        handle_weight_tying(model)
    if verbose:
        report_loading_status(model, result, converted)
    return model
