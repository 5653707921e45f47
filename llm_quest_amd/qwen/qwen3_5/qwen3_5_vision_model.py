"""Qwen3.5 vision tower on HIP kernels -- API of ``llm_quest/qwen/qwen3_5/qwen3_5_vision_model.py`` (BASELINE config 5).

Same class names, constructor-dict keys and state_dict keys as the reference (``patch_embed.conv_proj``, ``pos_embed``,
``blocks.N.{norm1,norm2,att.qkv,att.proj,ffn.lin1,ffn.lin2}``, ``merge_adapter.{norm,lin1,lin2}``).  Dtype flow as for the
ViT (vit_train.py): fp32 master parameters and residual stream, bf16 MFMA operands.  The whole tower is one autograd
node; forward and backward are HIP kernels only:

  Conv3d patches  -> coalesced 3-D im2row gather (bit-exact index map) + MFMA GEMM, learned pos-emb added per frame
  block           -> nn.LayerNorm kernel (eps inside sqrt) -> fused qkv GEMM(+bias) -> 2-D axial RoPE on q,k (the fused
                     QK kernel in RoPE-only mode, table row = patch index within the frame) -> flash attention (full,
                     D = 64) -> proj GEMM (+bias +residual) -> LayerNorm -> GEMM -> tanh-GELU kernel -> GEMM (+residual)
  merge adapter   -> LayerNorm -> m x m spatial merge (bit-exact row permutation kernel) -> GEMM -> erf-GELU -> GEMM
"""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd import ops
from llm_quest_amd.multimodal.vision_transformer.vit_attention import bf16_cached
from llm_quest_amd.multimodal.vision_transformer.vit_train import _acc, _bgrad, _wgrad

BF16, F32 = torch.bfloat16, torch.float32


def compute_angles_2d(base, head_dim, height_patches, width_patches, num_frames=1, dtype=torch.float32):
    """VisionRoPE.compute_angles_2d (common/rope.py:400-482): axial table cat([row*theta, col*theta]) duplicated."""
    assert head_dim % 4 == 0, "head_dim must be divisible by 4 for 2D RoPE"
    half = head_dim // 2
    theta = 1.0 / (base ** (2 * torch.arange(0, half // 2, dtype=dtype) / half))
    rows = torch.arange(height_patches, dtype=dtype).repeat_interleave(width_patches)
    cols = torch.arange(width_patches, dtype=dtype).repeat(height_patches)
    ang = torch.cat([torch.outer(rows, theta), torch.outer(cols, theta)], dim=-1)
    if num_frames > 1:
        ang = ang.repeat(num_frames, 1)
    ang = torch.cat([ang, ang], dim=-1)
    return torch.cos(ang), torch.sin(ang)


class PatchEmbedding3D(nn.Module):
    def __init__(self, img_width, img_height, num_channels, emb_dim, patch_size, temporal_patch_size):
        super().__init__()
        assert img_width % patch_size == 0, f"Image width {img_width} not divisible by patch size {patch_size}"
        assert img_height % patch_size == 0, f"Image height {img_height} not divisible by patch size {patch_size}"
        self.img_width, self.img_height = img_width, img_height
        self.patch_size, self.temporal_patch_size = patch_size, temporal_patch_size
        self.num_patches_per_image = (img_width * img_height) // patch_size**2
        ks = (temporal_patch_size, patch_size, patch_size)
        self.conv_proj = nn.Conv3d(num_channels, emb_dim, kernel_size=ks, stride=ks, padding=0, bias=True)  # never called: im2row + GEMM


class Qwen3_5VisionFFN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.lin1 = nn.Linear(cfg["vision_emb_dim"], cfg["vision_hidden_dim"])
        self.lin2 = nn.Linear(cfg["vision_hidden_dim"], cfg["vision_emb_dim"])


class Qwen3_5VisionAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.d_in = cfg["vision_emb_dim"]
        self.num_heads = cfg["vision_num_heads"]
        self.head_dim = self.d_in // self.num_heads
        self.qkv = nn.Linear(self.d_in, self.d_in * 3, bias=True)
        self.proj = nn.Linear(self.d_in, self.d_in, bias=True)


class Qwen3_5VisionTransformerBlock(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.norm1 = nn.LayerNorm(cfg["vision_emb_dim"], eps=1e-6)
        self.norm2 = nn.LayerNorm(cfg["vision_emb_dim"], eps=1e-6)
        self.att = Qwen3_5VisionAttention(cfg)
        self.ffn = Qwen3_5VisionFFN(cfg)


class ViTMergeAdapter(nn.Module):
    def __init__(self, vit_d_out, llm_d_in, n_height_patches, n_width_patches, spatial_merge_size=2):
        super().__init__()
        self.m = spatial_merge_size
        self.n_h_patches, self.n_w_patches = n_height_patches, n_width_patches
        self.merged_size = vit_d_out * self.m**2
        self.norm = nn.LayerNorm(vit_d_out, eps=1e-6)
        self.lin1 = nn.Linear(self.merged_size, self.merged_size)
        self.lin2 = nn.Linear(self.merged_size, llm_d_in)


# ------------------------------------------------------------------------------------------- kernels glue
def _ln(norm, x2d, out_dtype):
    return K.layernorm_fwd(x2d, norm.weight.detach(), norm.bias.detach(), out_dtype=out_dtype, eps=norm.eps, want_stats=True, mode=1)


def _ln_bwd(norm, x2d, mean, rsig, dy, dres):
    dx, dsc, dsh = K.layernorm_bwd(x2d, norm.weight.detach(), mean, rsig, dy, dres=dres, eps=norm.eps, mode=1)
    _acc(norm.weight, dsc.contiguous())
    _acc(norm.bias, dsh.contiguous())
    return dx


def _forward(m, pixels):
    pe = m.patch_embed
    if pixels.dim() != 5:
        raise ValueError("Qwen3_5VisionModel expects (b, c, t, h, w) pixels")
    B, C, T, H, W = pixels.shape
    assert H == pe.img_height and W == pe.img_width, f"Input image shape {tuple(pixels.shape)} does not match {pe.img_height}x{pe.img_width}"
    assert T % pe.temporal_patch_size == 0, f"time {T} is not divisible by temporal_patch_size {pe.temporal_patch_size}"
    nsp, frames = m.n_spatial_patches, T // pe.temporal_patch_size
    S, d = frames * nsp, m.emb_dim
    H_, Dh = m.num_heads, m.emb_dim // m.num_heads
    rows = K.patchify3d(pixels.contiguous().to(F32), pe.patch_size, pe.temporal_patch_size, out_dtype=BF16)
    wconv = bf16_cached(pe, "wconv", [pe.conv_proj.weight])
    # learned positional embedding of the patch's spatial index, the same for every frame: fused as the GEMM residual
    pos = m.pos_embed.weight.detach()[:nsp].contiguous()
    posr = pos.repeat(B * frames, 1)
    x = K.gemm(L.GEMM_NT, rows, wconv, bias=pe.conv_proj.bias.detach(), residual=posr, out_dtype=F32)
    tok_pos = torch.arange(nsp, dtype=torch.int32, device=x.device).repeat(B * frames)  # RoPE table row of every token
    cos, sin = m.cos, m.sin
    saved = []
    for blk in m.blocks:
        h1, mean1, rsig1 = _ln(blk.norm1, x, BF16)
        wqkv = bf16_cached(blk.att, "wqkv", [blk.att.qkv.weight])
        qkv = K.gemm(L.GEMM_NT, h1, wqkv, bias=blk.att.qkv.bias.detach())
        q, k, _ = K.qknorm_rope_fwd(qkv, None, None, cos, sin, tok_pos, H_, H_, Dh)
        ctx, lse = K.attn_fwd(q, k, qkv[:, 2 * d :], B, S, H_, H_, Dh, key_mask=None, causal=False, scale=Dh**-0.5)
        wo = bf16_cached(blk.att, "wo", [blk.att.proj.weight])
        x2 = K.gemm(L.GEMM_NT, ctx, wo, bias=blk.att.proj.bias.detach(), residual=x, out_dtype=F32)
        h2, mean2, rsig2 = _ln(blk.norm2, x2, BF16)
        y1, f = K.gemm_gelu_dual(h2, bf16_cached(blk.ffn, "w1", [blk.ffn.lin1.weight]), bias=blk.ffn.lin1.bias.detach(), tanh=True)
        x3 = K.gemm(L.GEMM_NT, f, bf16_cached(blk.ffn, "w2", [blk.ffn.lin2.weight]), bias=blk.ffn.lin2.bias.detach(), residual=x2, out_dtype=F32)
        saved.append((x, mean1, rsig1, h1, qkv, q, k, ctx, lse, x2, mean2, rsig2, h2, y1, f))
        x = x3
    ma = m.merge_adapter
    hn, meanm, rsigm = _ln(ma.norm, x, BF16)
    merged = K.merge_patches(hn, B * frames, ma.n_h_patches, ma.n_w_patches, ma.m)
    z1, a = K.gemm_gelu_dual(merged, bf16_cached(ma, "w1", [ma.lin1.weight]), bias=ma.lin1.bias.detach())
    out = K.gemm(L.GEMM_NT, a, bf16_cached(ma, "w2", [ma.lin2.weight]), bias=ma.lin2.bias.detach(), out_dtype=F32)
    n_merged = frames * (ma.n_h_patches // ma.m) * (ma.n_w_patches // ma.m)
    return out.view(B, n_merged, -1), (rows, tok_pos, saved, (x, meanm, rsigm, merged, z1, a), (B, frames, S, d))


def _backward(m, saved_all, dout):
    rows, tok_pos, saved, (xl, meanm, rsigm, merged, z1, a), (B, frames, S, d) = saved_all
    H_, Dh = m.num_heads, m.emb_dim // m.num_heads
    ma = m.merge_adapter
    g = dout.reshape(-1, dout.shape[-1]).contiguous()
    gb = g if g.dtype == BF16 else K.cast(g, BF16)
    dz1 = K.gemm_dgrad_gelu_bwd(gb, bf16_cached(ma, "w2", [ma.lin2.weight]), z1)
    _wgrad(ma.lin2.weight, gb, a)
    _bgrad(ma.lin2.bias, gb)
    dmerged = K.dgrad(dz1, bf16_cached(ma, "w1", [ma.lin1.weight]))
    _wgrad(ma.lin1.weight, dz1, merged)
    _bgrad(ma.lin1.bias, dz1)
    dhn = K.merge_patches(dmerged, B * frames, ma.n_h_patches, ma.n_w_patches, ma.m, inverse=True)
    dx = _ln_bwd(ma.norm, xl, meanm, rsigm, dhn, None)
    for blk, sv in zip(reversed(m.blocks), reversed(saved)):
        x, mean1, rsig1, h1, qkv, q, k, ctx, lse, x2, mean2, rsig2, h2, y1, f = sv
        wg = []  # this block's four weight gradients, one grouped launch
        dx3b = K.cast(dx, BF16)
        dy1 = K.gemm_dgrad_gelu_bwd(dx3b, bf16_cached(blk.ffn, "w2", [blk.ffn.lin2.weight]), y1, tanh=True)
        _wgrad(blk.ffn.lin2.weight, dx3b, f, wg)
        _bgrad(blk.ffn.lin2.bias, dx3b)
        dh2 = K.dgrad(dy1, bf16_cached(blk.ffn, "w1", [blk.ffn.lin1.weight]))
        _wgrad(blk.ffn.lin1.weight, dy1, h2, wg)
        _bgrad(blk.ffn.lin1.bias, dy1)
        dx2 = _ln_bwd(blk.norm2, x2, mean2, rsig2, dh2, dx)
        dx2b = K.cast(dx2, BF16)
        dctx = K.dgrad(dx2b, bf16_cached(blk.att, "wo", [blk.att.proj.weight]))
        _wgrad(blk.att.proj.weight, dx2b, ctx, wg)
        _bgrad(blk.att.proj.bias, dx2b)
        dqkv = torch.empty_like(qkv)
        dq, dk = torch.empty_like(q), torch.empty_like(k)
        K.attn_bwd(q, k, qkv[:, 2 * d :], ctx, dctx, lse, B, S, H_, H_, Dh, dq, dk, dqkv[:, 2 * d :], key_mask=None, causal=False, scale=Dh**-0.5)
        K.qknorm_rope_bwd(qkv, None, None, m.cos, m.sin, tok_pos, None, dq, dk, dqkv, H_, H_, Dh)  # RoPE^T only
        dh1 = K.dgrad(dqkv, bf16_cached(blk.att, "wqkv", [blk.att.qkv.weight]))
        _wgrad(blk.att.qkv.weight, dqkv, h1, wg)
        _bgrad(blk.att.qkv.bias, dqkv)
        dx = _ln_bwd(blk.norm1, x, mean1, rsig1, dh1, dx2)
        ops._flush_wgrads(wg)
    # patch projection + positional embedding (summed over batch and frames)
    pe = m.patch_embed
    nsp = m.n_spatial_patches
    if m.pos_embed.weight.requires_grad:
        gpos = K.colsum(dx.view(B * frames, nsp * d))
        full = torch.zeros_like(m.pos_embed.weight)
        K.copy2d(gpos.view(nsp, d), full[:nsp])
        _acc(m.pos_embed.weight, full)
    dxb = K.cast(dx, BF16)
    _wgrad(pe.conv_proj.weight, dxb, rows)
    _bgrad(pe.conv_proj.bias, dxb)


class _VisionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pixels, model, keep, *params):
        out, saved = _forward(model, pixels)
        ctx.model, ctx.saved = model, saved if keep else None
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("Qwen3_5VisionModel: backward through a forward that ran without grad mode")
        _backward(ctx.model, ctx.saved, dout)
        ctx.saved = None
        return (None, None, None) + (None,) * len(ctx.model._param_list)


class Qwen3_5VisionModel(nn.Module):
    """(b, c, t, h, w) pixels -> (b, t/tp * (gh/m) * (gw/m), llm_d_in) vision embeddings (reference: :241-370)."""

    def __init__(self, cfg):
        super().__init__()
        self.emb_dim, self.num_heads = cfg["vision_emb_dim"], cfg["vision_num_heads"]
        p = cfg["patch_size"]
        assert cfg["img_width"] % p == 0 and cfg["img_height"] % p == 0, "image size not divisible by the patch size"
        self.n_width_patches, self.n_height_patches = cfg["img_width"] // p, cfg["img_height"] // p
        self.n_spatial_patches = self.n_width_patches * self.n_height_patches
        assert self.n_spatial_patches <= cfg["num_position_embeddings"], "image too large for num_position_embeddings"
        self.patch_embed = PatchEmbedding3D(cfg["img_width"], cfg["img_height"], cfg["in_channels"], self.emb_dim, p, cfg["temporal_patch_size"])
        self.pos_embed = nn.Embedding(cfg["num_position_embeddings"], self.emb_dim)
        cos, sin = compute_angles_2d(cfg.get("vision_rope_base", 10_000), self.emb_dim // self.num_heads, self.n_height_patches, self.n_width_patches)
        self.register_buffer("cos", cos, persistent=False)
        self.register_buffer("sin", sin, persistent=False)
        self.blocks = nn.ModuleList([Qwen3_5VisionTransformerBlock(cfg) for _ in range(cfg["vision_n_layers"])])
        self.merge_adapter = ViTMergeAdapter(self.emb_dim, cfg["llm_d_in"], self.n_height_patches, self.n_width_patches, cfg["spatial_merge_size"])

    def forward(self, x):
        L.require_gpu(x)
        if not hasattr(self, "_param_list"):
            object.__setattr__(self, "_param_list", list(self.parameters()))
        keep = torch.is_grad_enabled() and any(p.requires_grad for p in self._param_list)
        return _VisionFn.apply(x, self, keep, *self._param_list)
