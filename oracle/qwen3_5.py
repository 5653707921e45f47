"""CPU restatement of the Qwen3.5 multimodal pieces (BASELINE config 5): vision tower, merge adapter, wrapper index ops.
TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The hybrid text stack (gated delta net, gated attention) is not
restated yet -- DESIGN.md lists it under "next"."""

import numpy as np
import torch
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- tables / index maps
def vision_rope_tables_2d(base, head_dim, gh, gw):
    """VisionRoPE.compute_angles_2d, num_frames=1 (common/rope.py:400-482): theta over half_dim//2 frequencies,
    angles = cat([row*theta, col*theta]) duplicated -> (gh*gw, head_dim), patches row-major."""
    half = head_dim // 2
    theta = 1.0 / (base ** (2 * torch.arange(0, half // 2, dtype=torch.float32) / half))
    rows = torch.arange(gh, dtype=torch.float32).repeat_interleave(gw)
    cols = torch.arange(gw, dtype=torch.float32).repeat(gh)
    ang = torch.cat([torch.outer(rows, theta), torch.outer(cols, theta)], dim=-1)
    ang = torch.cat([ang, ang], dim=-1)
    return torch.cos(ang), torch.sin(ang)


def patch3d_gather_index(channels, frames, height, width, patch, tpatch):
    """Source offsets of the Conv3d(k=s=(tp,P,P)) im2row matrix (qwen3_5_vision_model.py:79-107): tokens ordered
    (t', ph, pw), K ordered (c, dt, i, j).  int64 (tokens, C*tp*P*P) offsets into one (C,T,H,W) clip."""
    gt, gh, gw = frames // tpatch, height // patch, width // patch
    idx = np.empty((gt * gh * gw, channels * tpatch * patch * patch), dtype=np.int64)
    for tq in range(gt):
        for ph in range(gh):
            for pw in range(gw):
                row = (tq * gh + ph) * gw + pw
                k = 0
                for c in range(channels):
                    for dt in range(tpatch):
                        for i in range(patch):
                            base = ((c * frames + tq * tpatch + dt) * height + ph * patch + i) * width + pw * patch
                            idx[row, k : k + patch] = base + np.arange(patch)
                            k += patch
    return idx


def merge_row_source(frames, gh, gw, m):
    """ViTMergeAdapter view/permute (qwen3_5_vision_model.py:421-424): for merged token (f, bh, bw) the m*m source patch
    rows in (i, j) order.  int64 (frames*(gh/m)*(gw/m), m*m)."""
    out = np.empty((frames * (gh // m) * (gw // m), m * m), dtype=np.int64)
    r = 0
    for f in range(frames):
        for bh in range(gh // m):
            for bw in range(gw // m):
                out[r] = [(f * gh + bh * m + i) * gw + bw * m + j for i in range(m) for j in range(m)]
                r += 1
    return out


def position_ids_3d(input_ids, feeds_thw, image_token_id, merge):
    """Qwen3_5VLM.compute_3d_position_ids (qwen3_5_vlm_model.py:85-176) in numpy: text tokens advance all three axes by
    one; the image tokens of a feed share T = start (+frame), H/W = start + row/col of the MERGED grid; the next text
    token resumes at start + max(t, h/merge, w/merge).  Returns int64 (3, B, S)."""
    ids = np.asarray(input_ids)
    b, s = ids.shape
    if feeds_thw is None:
        return np.broadcast_to(np.arange(s, dtype=np.int64), (3, b, s)).copy()
    mask = ids == image_token_id
    inc = (~mask).astype(np.int64)
    local = np.zeros((3, b, s), dtype=np.int64)
    for bi in range(b):
        where = np.nonzero(mask[bi])[0]
        pos = 0
        for t, h, w in np.asarray(feeds_thw).tolist():
            mh, mw = h // merge, w // merge
            n = t * mh * mw
            if pos + n > len(where):
                break
            cur = where[pos : pos + n]
            inc[bi, cur[-1]] = max(t, mh, mw)
            li = np.arange(n)
            local[0, bi, cur] = li // (mh * mw)
            local[1, bi, cur] = (li % (mh * mw)) // mw
            local[2, bi, cur] = (li % (mh * mw)) % mw
            pos += n
    glob = np.cumsum(inc, axis=1) - inc
    return glob[None] + local


def masked_scatter_rows(emb, image_mask, vision_rows):
    """inputs_embs.masked_scatter(mask[..., None].expand_as(...), vision) (qwen3_5_vlm_model.py:209-211): row-major fill."""
    out = emb.clone()
    flat = out.view(-1, out.shape[-1])
    where = torch.nonzero(image_mask.reshape(-1)).squeeze(1)
    flat[where] = vision_rows.reshape(-1, out.shape[-1])[: len(where)].to(out.dtype)
    return out


# --------------------------------------------------------------------------- vision tower
def patch_embed_3d(x, w, b, patch, tpatch):
    bsz, c, t, h, wd = x.shape
    gt, gh, gw = t // tpatch, h // patch, wd // patch
    rows = x.reshape(bsz, c, gt, tpatch, gh, patch, gw, patch).permute(0, 2, 4, 6, 1, 3, 5, 7).reshape(bsz, gt * gh * gw, -1)
    return rows @ w.reshape(w.shape[0], -1).t() + b


def vision35_forward(sd, cfg, pixels):
    """Qwen3_5VisionModel.forward (qwen3_5_vision_model.py:336-370): Conv3d patches + learned pos-emb (per frame) ->
    blocks {LayerNorm(1e-6) -> fused qkv(+bias) -> 2-D RoPE on q,k -> full attention -> proj ; LayerNorm -> tanh-GELU FFN}
    -> ViTMergeAdapter {LayerNorm -> m x m merge -> Linear -> GELU(erf) -> Linear}."""
    d, nh = cfg["vision_emb_dim"], cfg["vision_num_heads"]
    dh = d // nh
    gh, gw = cfg["img_height"] // cfg["patch_size"], cfg["img_width"] // cfg["patch_size"]
    x = patch_embed_3d(pixels, sd["patch_embed.conv_proj.weight"], sd["patch_embed.conv_proj.bias"], cfg["patch_size"], cfg["temporal_patch_size"])
    b, s, _ = x.shape
    frames = s // (gh * gw)
    x = x + sd["pos_embed.weight"][: gh * gw].repeat(frames, 1)
    cos, sin = vision_rope_tables_2d(cfg["vision_rope_base"], dh, gh, gw)
    cos, sin = cos.repeat(frames, 1), sin.repeat(frames, 1)
    for i in range(cfg["vision_n_layers"]):
        p = f"blocks.{i}."
        h = F.layer_norm(x, (d,), sd[p + "norm1.weight"], sd[p + "norm1.bias"], 1e-6)
        q, k, v = F.linear(h, sd[p + "att.qkv.weight"], sd[p + "att.qkv.bias"]).chunk(3, dim=-1)
        q, k, v = (t.view(b, s, nh, dh).transpose(1, 2) for t in (q, k, v))
        q, k = ops.rope_apply(q, cos, sin), ops.rope_apply(k, cos, sin)
        ctx = ops.full_attention_core(q, k, v).transpose(1, 2).contiguous().view(b, s, d)
        x = F.linear(ctx, sd[p + "att.proj.weight"], sd[p + "att.proj.bias"]) + x
        h = F.layer_norm(x, (d,), sd[p + "norm2.weight"], sd[p + "norm2.bias"], 1e-6)
        h = F.gelu(F.linear(h, sd[p + "ffn.lin1.weight"], sd[p + "ffn.lin1.bias"]), approximate="tanh")
        x = F.linear(h, sd[p + "ffn.lin2.weight"], sd[p + "ffn.lin2.bias"]) + x
    m = cfg["spatial_merge_size"]
    h = F.layer_norm(x, (d,), sd["merge_adapter.norm.weight"], sd["merge_adapter.norm.bias"], 1e-6)
    src = torch.from_numpy(merge_row_source(frames, gh, gw, m))
    h = h[:, src.reshape(-1)].reshape(b, src.shape[0], m * m * d)
    h = F.gelu(F.linear(h, sd["merge_adapter.lin1.weight"], sd["merge_adapter.lin1.bias"]))
    return F.linear(h, sd["merge_adapter.lin2.weight"], sd["merge_adapter.lin2.bias"])


# --------------------------------------------------------------------------- the wrapper
def vlm35_forward(sd, cfg, input_ids, pixels=None, attn_mask=None):
    """Qwen3_5VLM.forward (qwen3_5_vlm_model.py:178-227) over one state dict with the reference's key names (``vision_model.*``,
    ``language_model.*``): token embedding -> vision tower -> masked_scatter at the placeholder positions (vision rows cast to the text
    dtype) -> 3-D position ids -> hybrid text model.  Returns (logits, position_ids)."""
    from . import qwen3_5_text as q35t

    txt = {k[len("language_model."):]: v for k, v in sd.items() if k.startswith("language_model.")}
    emb = torch.nn.functional.embedding(input_ids, txt["emb_dict.weight"])
    feeds = None
    if pixels is not None:
        vis = vision35_forward({k[len("vision_model."):]: v for k, v in sd.items() if k.startswith("vision_model.")}, cfg, pixels)
        emb = masked_scatter_rows(emb, input_ids == cfg["image_token_id"], vis)
        t = pixels.shape[2] // cfg["temporal_patch_size"]
        feeds = [[t, cfg["img_height"] // cfg["patch_size"], cfg["img_width"] // cfg["patch_size"]]] * 1
    pid = torch.from_numpy(position_ids_3d(input_ids.numpy(), feeds, cfg["image_token_id"], cfg["spatial_merge_size"]))
    return q35t.text_model_forward(txt, cfg, attn_mask=attn_mask, inputs_embs=emb, position_ids=pid), pid
