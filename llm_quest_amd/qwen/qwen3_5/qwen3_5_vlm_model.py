"""Qwen3.5 multimodal wrapper pieces on HIP kernels -- API of ``llm_quest/qwen/qwen3_5/qwen3_5_vlm_model.py``.

BASELINE config 5: the vision tower, the masked-scatter early fusion, the 3-D MRoPE position
ids and the hybrid text stack (``Qwen3_5TextModel``: gated delta net + gated attention) all run on HIP kernels.  ``language_model``
may still be passed in (anything with ``emb_dict`` and the reference's ``forward(inputs_embs=..., position_ids=..., attn_mask=...)``).
"""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd.qwen.qwen3_5.qwen3_5_vision_model import Qwen3_5VisionModel


class _ScatterFn(torch.autograd.Function):
    """inputs_embs.masked_scatter(image_mask[..., None], vision_embeds): row-major fill, bit-exact (reference :209-211)."""

    @staticmethod
    def forward(ctx, emb, vis, mask_u8, slot):
        B, S, d = emb.shape
        vis_dtype, vis_shape = vis.dtype, vis.shape
        vis = K.cast(vis.reshape(-1, d).contiguous(), emb.dtype)  # vision_embeds.to(inputs_embs.dtype)
        out = K.scatter_rows_fwd(emb.reshape(B * S, d).contiguous(), vis, mask_u8, slot)
        ctx.meta = (B, S, d, vis.shape, vis_dtype, mask_u8, slot)
        ctx.vis_shape = vis_shape
        return out.view(B, S, d)

    @staticmethod
    def backward(ctx, g):
        B, S, d, vshape, vis_dtype, mask_u8, slot = ctx.meta
        d_emb, d_vis = K.scatter_rows_bwd(g.reshape(B * S, d), vshape[0], mask_u8, slot)
        return d_emb.view(B, S, d), K.cast(d_vis, vis_dtype).view(ctx.vis_shape), None, None


def fuse_vision_embeddings(inputs_embs, image_mask, vision_embeds):
    """Replace the rows at image-placeholder positions (row-major over (b, s)) by the vision rows, cast to the text dtype."""
    L.require_gpu(inputs_embs, vision_embeds, image_mask)
    mask_flat = image_mask.reshape(-1)
    n_slots = vision_embeds.numel() // vision_embeds.shape[-1]
    # integer metadata (as upstream computes it on the host): slot[t] = number of placeholders before t
    slot = (torch.cumsum(mask_flat.to(torch.int32), 0, dtype=torch.int32) - mask_flat.to(torch.int32)).contiguous()
    slot = torch.clamp(slot, max=max(n_slots - 1, 0))  # more placeholders than vision rows is an upstream error; never read out of range
    return _ScatterFn.apply(inputs_embs, vision_embeds, mask_flat.to(torch.uint8).contiguous(), slot)


class Qwen3_5VLM(nn.Module):
    def __init__(self, cfg, language_model=None):
        super().__init__()
        self.image_token_id = cfg.get("image_token_id", 248056)
        self.merge_size = cfg["spatial_merge_size"]
        self.cfg = cfg
        self.vision_model = Qwen3_5VisionModel(cfg)
        if language_model is None:
            from llm_quest_amd.qwen.qwen3_5.qwen3_5_text_model import Qwen3_5TextModel

            language_model = Qwen3_5TextModel(cfg)
        self.language_model = language_model

    def get_feeds_3d_shape(self, image_pixels):
        """(1, 3) tensor [t', gh, gw] in patches for the single visual feed (reference :46-83)."""
        gh, gw = self.vision_model.n_height_patches, self.vision_model.n_width_patches
        if image_pixels.dim() == 5:
            frames = image_pixels.shape[2] // self.cfg["temporal_patch_size"]
        else:
            frames = image_pixels.shape[1] // (gh * gw)
        return torch.tensor([[frames, gh, gw]])

    def compute_3d_position_ids(self, input_ids, feeds_3d_shape=None, image_mask=None):
        """(3, b, s) MRoPE position ids (reference :85-176).  Integer index logic on whatever device ``input_ids`` lives."""
        b, s = input_ids.shape
        dev = input_ids.device
        if feeds_3d_shape is None:
            return torch.arange(s, device=dev).view(1, 1, -1).expand(3, b, s)
        if image_mask is None:
            image_mask = input_ids == self.image_token_id
        inc = (~image_mask).long()
        local = torch.zeros(3, b, s, device=dev, dtype=torch.long)
        feeds = [tuple(int(v) for v in row) for row in feeds_3d_shape.tolist()]
        for bi in range(b):
            where = torch.where(image_mask[bi])[0]
            pos = 0
            for t, h, w in feeds:
                mh, mw = h // self.merge_size, w // self.merge_size
                n = t * mh * mw
                if pos + n > len(where):
                    break
                cur = where[pos : pos + n]
                inc[bi, cur[-1]] = max(t, mh, mw)
                li = torch.arange(n, device=dev)
                local[0, bi, cur] = li // (mh * mw)
                local[1, bi, cur] = (li % (mh * mw)) // mw
                local[2, bi, cur] = (li % (mh * mw)) % mw
                pos += n
        glob = torch.cumsum(inc, dim=1) - inc
        return glob.unsqueeze(0) + local

    def forward(self, input_ids, image_pixels=None, feeds_3d_shape=None, attn_mask=None):
        inputs_embs = self.language_model.emb_dict(input_ids)
        image_mask = None
        if image_pixels is not None:
            vision_embeds = self.vision_model(image_pixels)
            image_mask = input_ids == self.image_token_id
            inputs_embs = fuse_vision_embeddings(inputs_embs, image_mask, vision_embeds)
            feeds_3d_shape = self.get_feeds_3d_shape(image_pixels)
        position_ids = self.compute_3d_position_ids(input_ids, feeds_3d_shape, image_mask=image_mask)
        return self.language_model(inputs_embs=inputs_embs, position_ids=position_ids, attn_mask=attn_mask)
