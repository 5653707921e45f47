"""Model configuration dicts (the constructor-dict keys are part of the API) -- subset of the reference's ``config.py``
covering the hot path: GPT-2 (config 1), ViT-Base (config 2/4) and dense Qwen3 (config 3/4)."""

import torch


def _pick_device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


auto_device = _pick_device()

GPT_CONFIG_124M = {"vocab_size": 50257, "context_length": 1024, "emb_dim": 768, "n_heads": 12, "n_layers": 12, "drop_rate": 0.0, "qkv_bias": True}

VIT_BASE_CONFIG = {
    "img_width": 224, "img_height": 224, "patch_size": 16, "num_channels": 3, "emb_dim": 768, "n_layers": 12, "n_heads": 12,
    "drop_rate": 0.1, "qkv_bias": True, "num_classes": 100,
}

_GPT2_SIZES = {
    "gpt_s": {"emb_dim": 768, "n_layers": 12, "n_heads": 12},
    "gpt_m": {"emb_dim": 1024, "n_layers": 24, "n_heads": 16},
    "gpt_l": {"emb_dim": 1280, "n_layers": 36, "n_heads": 20},
    "gpt_xl": {"emb_dim": 1600, "n_layers": 48, "n_heads": 25},
}


def gpt2_config_creator(gpt_size):
    cfg = dict(GPT_CONFIG_124M)
    cfg.update(_GPT2_SIZES[gpt_size], qkv_bias=True)
    return cfg


_QWEN3_DENSE = {
    "0.6B": dict(emb_dim=1024, n_layers=28, n_heads=16, num_kv_groups=8, hidden_dim=3072, context_length=40_960, tie_embeddings=True),
    "1.7B": dict(emb_dim=2048, n_layers=28, n_heads=16, num_kv_groups=8, hidden_dim=6144, context_length=40_960, tie_embeddings=True),
    "4B": dict(emb_dim=2560, n_layers=36, n_heads=32, num_kv_groups=8, hidden_dim=9728, context_length=40_960, tie_embeddings=True),
}


def qwen3_config_creator(model_size="0.6B", base_model=True):
    """Dense Qwen3 configs (vocab 151 936, RoPE base 1e6, head_dim 128, bf16).  Override ``context_length`` for training
    runs: the reference registers a dense (ctx, ctx) mask buffer, 1.68 GB at the stock 40 960."""
    if model_size not in _QWEN3_DENSE:
        raise KeyError(f"only the dense Qwen3 sizes {sorted(_QWEN3_DENSE)} are on the hot path (MoE is out of scope)")
    cfg = {
        "vocab_size": 151_936, "rope_base": 1_000_000, "head_dim": 128, "dtype": torch.bfloat16, "model_type": "dense",
        "model_path": f"Qwen/Qwen3-{model_size}{'-Base' if base_model else ''}",
    }
    cfg.update(_QWEN3_DENSE[model_size])
    return cfg


# Qwen3.5-0.8B, vision + text (reference config.py:361-416; keys are the constructor API of Qwen3_5VLM / Qwen3_5TextModel)
QWEN3_5_08B_CONFIG = {
    "model_path": "Qwen/Qwen3.5-0.8B",
    "vocab_size": 248_320, "emb_dim": 1024, "hidden_dim": 3584, "n_layers": 24, "linear_sdpa_ratio": 4, "n_heads": 8, "num_kv_groups": 2,
    "head_dim": 256, "rope_base": 10_000_000, "partial_rope_factor": 0.25, "context_length": 8192,
    "linear_num_qk_heads": 16, "linear_num_value_heads": 16, "linear_qk_head_dim": 128, "linear_value_head_dim": 128,
    "linear_conv_kernel_size": 4, "tie_embeddings": True, "dtype": torch.bfloat16, "p_dropout": 0.0, "training": False,
    "mrope_section": [11, 11, 10],
    "vision_n_layers": 12, "vision_emb_dim": 768, "vision_hidden_act": "gelu_pytorch_tanh", "vision_hidden_dim": 3072, "vision_num_heads": 12,
    "llm_d_in": 1024, "in_channels": 3, "patch_size": 16, "spatial_merge_size": 2, "temporal_patch_size": 2, "num_position_embeddings": 2304,
    "img_width": 384, "img_height": 384, "vision_rope_base": 10_000,
    "image_token_id": 248056, "vision_start_token_id": 248053, "vision_end_token_id": 248054, "video_token_id": 248057,
    "image_mean": [0.5, 0.5, 0.5], "image_std": [0.5, 0.5, 0.5],
}
