"""Integer / index restatements (numpy).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

These are the bit-exact parts of the path: which input element lands where.
"""

import numpy as np


def patch_gather_index(channels, height, width, patch):
    """Flat source index for every (patch, k) element of the im2row matrix.

    Follows ``nn.Conv2d(C, D, kernel=P, stride=P)`` + ``flatten(2).transpose(1, 2)``
    (multimodal/vision_transformer/vit_model.py:50-57, 77-84): patches are
    row-major over (ph, pw); inside a patch K is ordered (c, i, j), i.e. the
    Conv2d weight ``(D, C, P, P)`` flattened over its last three dims.

    Returns int64 array (num_patches, C*P*P) of offsets into one CHW image.
    """
    gh, gw = height // patch, width // patch
    idx = np.empty((gh * gw, channels * patch * patch), dtype=np.int64)
    for ph in range(gh):
        for pw in range(gw):
            row = ph * gw + pw
            k = 0
            for c in range(channels):
                for i in range(patch):
                    base = c * height * width + (ph * patch + i) * width + pw * patch
                    idx[row, k : k + patch] = base + np.arange(patch)
                    k += patch
    return idx


def early_fusion_row_source(batch, n_vision, n_text):
    """Row provenance of ``torch.cat([vision, text], dim=1)`` (multimodal/vlm_engine.py:111-114).

    Returns int64 (batch, n_vision + n_text, 2): [:, :, 0] is 0 for a vision row, 1 for a
    text row; [:, :, 1] is the row index inside that source for the same batch element.
    """
    s = n_vision + n_text
    out = np.empty((batch, s, 2), dtype=np.int64)
    out[:, :n_vision, 0] = 0
    out[:, :n_vision, 1] = np.arange(n_vision)
    out[:, n_vision:, 0] = 1
    out[:, n_vision:, 1] = np.arange(n_text)
    return out


def fused_attention_mask(text_mask, n_vision):
    """``cat([ones(B, S_v, bool), text_mask], 1)`` (multimodal/vlm_engine.py:116-119)."""
    text_mask = np.asarray(text_mask).astype(bool)
    b = text_mask.shape[0]
    return np.concatenate([np.ones((b, n_vision), dtype=bool), text_mask], axis=1)


def vlm_label_rows(input_ids, text_mask, n_vision):
    """Targets and logits-row indices used by ``vlm_loss`` (multimodal/vlm_engine.py:36-39).

    logits[:, n_vision-1 : -1] predicts input_ids; padded positions become -100.
    Returns (labels (B, T) int64 with -100 on pads, logits_rows (T,) int64).
    """
    ids = np.asarray(input_ids).astype(np.int64)
    m = np.asarray(text_mask)
    labels = np.where(m == 0, -100, ids)
    t = ids.shape[1]
    rows = np.arange(n_vision - 1, n_vision - 1 + t, dtype=np.int64)
    return labels, rows


def gqa_head_map(n_heads, n_kv_groups):
    """kv head serving each query head under ``repeat_interleave(rep, dim=1)``
    (qwen/qwen3/qwen3_attention.py:121-122): q-head h reads kv-head h // rep."""
    rep = n_heads // n_kv_groups
    return np.arange(n_heads, dtype=np.int64) // rep


def attention_visibility(seq_len, key_mask=None, causal=True):
    """Boolean (B or 1, S, S) "may attend" matrix = NOT(causal_upper | ~key_mask)
    (common/buffers.py:25-37 triu(…,1) True = masked; qwen3_attention.py:130-137)."""
    vis = np.ones((seq_len, seq_len), dtype=bool)
    if causal:
        vis = ~np.triu(np.ones((seq_len, seq_len), dtype=bool), k=1)
    vis = vis[None]
    if key_mask is not None:
        km = np.asarray(key_mask).astype(bool)
        vis = vis & km[:, None, :]
    return vis
