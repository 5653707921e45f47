"""Hugging Face -> this package's parameter names for Qwen3 (API of
``llm_quest/qwen/qwen3/qwen3_weight_loading.py`` and the helpers it takes from ``llm_quest/utils.py:956-1035``).

There is no network on the build / GPU boxes, so the download step is replaced by a local source: a dict of tensors or a
``.safetensors`` file / directory of shards that the caller already has.  Name mapping, shape checks, weight tying and the
loading report behave as upstream; the loaded parameters land in the model's arenas (``load_state_dict`` copies into the
existing views, so fused QKV / gate-up GEMMs see them without any re-packing).
"""

import glob
import os

import torch


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


def convert_weights(hf_state_dict, our_state_dict, remapping_rules, ignored_prefixes=None):
    """Rename HF tensors to our names; keep those whose target exists with the same shape (reference utils.py:956-1000).
    Returns ``converted`` (name -> tensor); mismatches are reported on stdout like upstream and skipped."""
    ignored_prefixes = tuple(ignored_prefixes or ())
    converted, skipped = {}, 0
    for hf_name, w in hf_state_dict.items():
        if ignored_prefixes and hf_name.startswith(ignored_prefixes):
            skipped += 1
            continue
        name = hf_name
        for pattern, replacement in remapping_rules:
            if pattern in name:
                name = name.replace(pattern, replacement)
                if pattern == hf_name:  # whole-name rule: done
                    break
        if name not in our_state_dict:
            print(f"WARNING: No match for HF weight '{hf_name}' -> tried '{name}'")
        elif w.shape != our_state_dict[name].shape:
            print(f"WARNING: Shape mismatch: {name}: HF {tuple(w.shape)} vs Ours {tuple(our_state_dict[name].shape)}")
        else:
            converted[name] = w.clone()
    if skipped:
        print(f"Skipped {skipped} weights")
    return converted


def handle_weight_tying(model):
    """Re-tie ``out_head.weight`` to ``emb_dict.weight`` after loading (reference utils.py:1003-1022)."""
    if not getattr(model, "tie_embeddings", False):
        return False
    if model.emb_dict.weight.shape != model.out_head.weight.shape:
        print(f"WARNING: Shape mismatch for weight tying: {tuple(model.emb_dict.weight.shape)} vs {tuple(model.out_head.weight.shape)}")
        return False
    model.out_head.weight = model.emb_dict.weight
    return model.out_head.weight is model.emb_dict.weight


def report_loading_status(model, load_result, converted_weights):
    print(f"Loaded {len(converted_weights)}/{len(model.state_dict())} weights successfully")
    if load_result.missing_keys:
        print(f"Missing keys ({len(load_result.missing_keys)}): {load_result.missing_keys}")
        print("-> out_head expected with tie_embeddings=True; buffers mask / cos / sin are rebuilt locally")
    if load_result.unexpected_keys:
        print(f"Unexpected keys: {load_result.unexpected_keys}")


def read_local_checkpoint(source):
    """dict of tensors, a ``.safetensors`` file, or a directory of ``*.safetensors`` shards -> one state dict."""
    if isinstance(source, dict):
        return source
    from safetensors.torch import load_file

    if os.path.isdir(source):
        files = sorted(glob.glob(os.path.join(source, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no .safetensors shards under {source}")
    else:
        files = [source]
    sd = {}
    for f in files:
        sd.update(load_file(f))
    return sd


def load_qwen3_weights(model, model_cfg, source=None, verbose=True):
    """Convert + load a Hugging Face Qwen3 checkpoint (reference qwen3_weight_loading.py:69-103).  ``source`` is required
    here: this package never opens a network connection."""
    if source is None:
        raise ValueError("load_qwen3_weights: pass a local checkpoint (dict, .safetensors file or shard directory); "
                         "downloading from the Hugging Face hub is not available in this environment")
    hf_state_dict = read_local_checkpoint(source)
    converted = convert_weights(hf_state_dict, model.state_dict(), get_remapping_rules(model_cfg))
    with torch.no_grad():
        result = model.load_state_dict(converted, strict=False)
        handle_weight_tying(model)
    if verbose:
        report_loading_status(model, result, converted)
    return model
