"""Qwen3.5 vision tower on HIP kernels -- API of ``llm_quest/qwen/qwen3_5/qwen3_5_vision_model.py`` (BASELINE config 5).

Same class names, constructor-dict keys, ``forward`` signatures and state_dict keys as the reference (``patch_embed.conv_proj``,
``pos_embed``, ``blocks.N.{norm1,norm2,att.qkv,att.proj,ffn.lin1,ffn.lin2}``, ``merge_adapter.{norm,lin1,lin2}``).  Dtype flow as for
the ViT (vit_train.py): fp32 master parameters and residual stream, bf16 MFMA operands.  Forward and backward are HIP kernels only:

  Conv3d patches  -> coalesced 3-D im2row gather (bit-exact index map) + MFMA GEMM, learned pos-emb added per frame
  block           -> nn.LayerNorm kernel (eps inside sqrt) -> fused qkv GEMM(+bias) -> 2-D axial RoPE on q,k (the fused
                     QK kernel in RoPE-only mode, table row = patch index within the frame) -> flash attention (full,
                     D = 64) -> proj GEMM (+bias +residual) -> LayerNorm -> GEMM -> tanh-GELU kernel -> GEMM (+residual)
  merge adapter   -> LayerNorm -> m x m spatial merge (bit-exact row permutation kernel) -> GEMM -> erf-GELU -> GEMM

The tower is written as pieces (``_patch_*``, ``_attn_*``, ``_ffn_*``, ``_block_*``, ``_merge_*``: a forward that returns what its
backward needs, and that backward).  ``Qwen3_5VisionModel.forward`` is ONE autograd node over all of them; every sub-module's own
``forward`` (the reference's entry points, qwen3_5_vision_model.py:88,124,153,218,411) is one autograd node over its piece.
"""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd import ops
from llm_quest_amd.common.rope import VisionRoPE
from llm_quest_amd.multimodal.vision_transformer.vit_attention import bf16_cached
from llm_quest_amd.multimodal.vision_transformer.vit_train import _acc, _bgrad, _wgrad

BF16, F32 = torch.bfloat16, torch.float32


# ------------------------------------------------------------------------------------------- kernels glue
def _ln(norm, x2d, out_dtype):
    return K.layernorm_fwd(x2d, norm.weight.detach(), norm.bias.detach(), out_dtype=out_dtype, eps=norm.eps, want_stats=True, mode=1)


def _ln_bwd(norm, x2d, mean, rsig, dy, dres):
    dx, dsc, dsh = K.layernorm_bwd(x2d, norm.weight.detach(), mean, rsig, dy, dres=dres, eps=norm.eps, mode=1)
    _acc(norm.weight, dsc.contiguous())
    _acc(norm.bias, dsh.contiguous())
    return dx


def _as_f32_rows(x):
    """(..., d) fp32 / bf16 -> contiguous fp32 [rows, d]."""
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    return x2 if x2.dtype == F32 else K.cast(x2, F32)


def _as_bf16_rows(x):
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    return x2 if x2.dtype == BF16 else K.cast(x2, BF16)


def _like(y2d, ref, last):
    """rows [n, last] in ref's dtype, shaped like ref's leading dimensions when the row count is ref's (else left 2-D)."""
    y = y2d if y2d.dtype == ref.dtype else K.cast(y2d, ref.dtype)
    lead = ref.shape[:-1]
    return y.view(*lead, last) if y.shape[0] == lead.numel() else y


def _rope_tables(cos, sin, S, Dh, device):
    if tuple(cos.shape) != tuple(sin.shape) or cos.shape[0] < S or cos.shape[1] != Dh:
        raise ValueError(f"vision attention: cos/sin must be (>= seq_len = {S}, head_dim = {Dh}), got {tuple(cos.shape)}")
    c = cos.to(device=device, dtype=F32)
    s = sin.to(device=device, dtype=F32)
    return (c if c.is_contiguous() else c.contiguous()), (s if s.is_contiguous() else s.contiguous())


_pos_cache = {}


def _token_rows(n_rows, repeats, device):
    """int32 [repeats * n_rows]: 0..n_rows-1 repeated (the RoPE table row of every token)."""
    key = (n_rows, repeats, str(device))
    t = _pos_cache.get(key)
    if t is None:
        t = torch.arange(n_rows, dtype=torch.int32, device=device).repeat(repeats)
        _pos_cache[key] = t
    return t


# ------------------------------------------------------------------------------------------- pieces
def _patch_fwd(pe, pixels, residual=None):
    if pixels.dim() != 5:
        raise ValueError("PatchEmbedding3D expects (b, c, t, h, w) pixels")
    B, C, T, H, W = pixels.shape
    assert H == pe.img_height and W == pe.img_width, f"Input image shape {pixels.shape} does not match expected shape {pe.img_height}x{pe.img_width}"
    assert T % pe.temporal_patch_size == 0, f"Input time shape {T} is not divisible by temporal_patch_size {pe.temporal_patch_size}"
    px = pixels if pixels.dtype == F32 else pixels.to(F32)
    rows = K.patchify3d(px.contiguous(), pe.patch_size, pe.temporal_patch_size, out_dtype=BF16)
    wconv = bf16_cached(pe, "wconv", [pe.conv_proj.weight])
    x = K.gemm(L.GEMM_NT, rows, wconv, bias=pe.conv_proj.bias.detach(), residual=residual, out_dtype=F32)
    return x, rows


def _patch_bwd(pe, rows, dx_f32):
    dxb = K.cast(dx_f32, BF16)
    _wgrad(pe.conv_proj.weight, dxb, rows)
    _bgrad(pe.conv_proj.bias, dxb)


def _attn_fwd(att, h1, B, S, cos, sin, tok_pos, residual=None, out_dtype=F32):
    """h1 bf16 [B*S, d] -> proj(attention) (+ residual); saved for the backward."""
    d, H_, Dh = att.d_in, att.num_heads, att.head_dim
    qkv = K.gemm(L.GEMM_NT, h1, bf16_cached(att, "wqkv", [att.qkv.weight]), bias=att.qkv.bias.detach())
    q, k, _ = K.qknorm_rope_fwd(qkv, None, None, cos, sin, tok_pos, H_, H_, Dh)
    ctx, lse = K.attn_fwd(q, k, qkv[:, 2 * d :], B, S, H_, H_, Dh, key_mask=None, causal=False, scale=Dh**-0.5)
    y = K.gemm(L.GEMM_NT, ctx, bf16_cached(att, "wo", [att.proj.weight]), bias=att.proj.bias.detach(), residual=residual, out_dtype=out_dtype)
    return y, (h1, qkv, q, k, ctx, lse, B, S, cos, sin, tok_pos)


def _attn_bwd(att, saved, dyb, wg=None):
    """dyb bf16 [B*S, d] = gradient of the projection output -> dh1 bf16; parameter gradients accumulate in .grad."""
    h1, qkv, q, k, ctx, lse, B, S, cos, sin, tok_pos = saved
    d, H_, Dh = att.d_in, att.num_heads, att.head_dim
    dctx = K.dgrad(dyb, bf16_cached(att, "wo", [att.proj.weight]))
    _wgrad(att.proj.weight, dyb, ctx, wg)
    _bgrad(att.proj.bias, dyb)
    dqkv = torch.empty_like(qkv)
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    K.attn_bwd(q, k, qkv[:, 2 * d :], ctx, dctx, lse, B, S, H_, H_, Dh, dq, dk, dqkv[:, 2 * d :], key_mask=None, causal=False, scale=Dh**-0.5)
    K.qknorm_rope_bwd(qkv, None, None, cos, sin, tok_pos, None, dq, dk, dqkv, H_, H_, Dh)  # RoPE^T only
    dh1 = K.dgrad(dqkv, bf16_cached(att, "wqkv", [att.qkv.weight]))
    _wgrad(att.qkv.weight, dqkv, h1, wg)
    _bgrad(att.qkv.bias, dqkv)
    return dh1


def _ffn_fwd(ffn, h2, residual=None, out_dtype=F32):
    y1, f = K.gemm_gelu_dual(h2, bf16_cached(ffn, "w1", [ffn.lin1.weight]), bias=ffn.lin1.bias.detach(), tanh=True)
    y = K.gemm(L.GEMM_NT, f, bf16_cached(ffn, "w2", [ffn.lin2.weight]), bias=ffn.lin2.bias.detach(), residual=residual, out_dtype=out_dtype)
    return y, (h2, y1, f)


def _ffn_bwd(ffn, saved, dyb, wg=None):
    h2, y1, f = saved
    dy1 = K.gemm_dgrad_gelu_bwd(dyb, bf16_cached(ffn, "w2", [ffn.lin2.weight]), y1, tanh=True)
    _wgrad(ffn.lin2.weight, dyb, f, wg)
    _bgrad(ffn.lin2.bias, dyb)
    dh2 = K.dgrad(dy1, bf16_cached(ffn, "w1", [ffn.lin1.weight]))
    _wgrad(ffn.lin1.weight, dy1, h2, wg)
    _bgrad(ffn.lin1.bias, dy1)
    return dh2


def _block_fwd(blk, x, B, S, cos, sin, tok_pos):
    """x fp32 [B*S, d] -> fp32 [B*S, d] (both residual adds are GEMM epilogues)."""
    h1, mean1, rsig1 = _ln(blk.norm1, x, BF16)
    x2, att_saved = _attn_fwd(blk.att, h1, B, S, cos, sin, tok_pos, residual=x)
    h2, mean2, rsig2 = _ln(blk.norm2, x2, BF16)
    x3, ffn_saved = _ffn_fwd(blk.ffn, h2, residual=x2)
    return x3, (x, mean1, rsig1, att_saved, x2, mean2, rsig2, ffn_saved)


def _block_bwd(blk, saved, dx):
    x, mean1, rsig1, att_saved, x2, mean2, rsig2, ffn_saved = saved
    wg = []  # this block's four weight gradients, one grouped launch
    dh2 = _ffn_bwd(blk.ffn, ffn_saved, K.cast(dx, BF16), wg)
    dx2 = _ln_bwd(blk.norm2, x2, mean2, rsig2, dh2, dx)
    dh1 = _attn_bwd(blk.att, att_saved, K.cast(dx2, BF16), wg)
    dx0 = _ln_bwd(blk.norm1, x, mean1, rsig1, dh1, dx2)
    ops._flush_wgrads(wg)
    return dx0


def _merge_fwd(ma, x, n_images):
    """x fp32 [n_images * gh * gw, d] -> fp32 [n_images * (gh/m) * (gw/m), llm_d_in]."""
    hn, meanm, rsigm = _ln(ma.norm, x, BF16)
    merged = K.merge_patches(hn, n_images, ma.n_h_patches, ma.n_w_patches, ma.m)
    z1, a = K.gemm_gelu_dual(merged, bf16_cached(ma, "w1", [ma.lin1.weight]), bias=ma.lin1.bias.detach())
    out = K.gemm(L.GEMM_NT, a, bf16_cached(ma, "w2", [ma.lin2.weight]), bias=ma.lin2.bias.detach(), out_dtype=F32)
    return out, (x, meanm, rsigm, merged, z1, a, n_images)


def _merge_bwd(ma, saved, gb):
    """gb bf16 [merged rows, llm_d_in] -> dx fp32 [rows, d]."""
    xl, meanm, rsigm, merged, z1, a, n_images = saved
    dz1 = K.gemm_dgrad_gelu_bwd(gb, bf16_cached(ma, "w2", [ma.lin2.weight]), z1)
    _wgrad(ma.lin2.weight, gb, a)
    _bgrad(ma.lin2.bias, gb)
    dmerged = K.dgrad(dz1, bf16_cached(ma, "w1", [ma.lin1.weight]))
    _wgrad(ma.lin1.weight, dz1, merged)
    _bgrad(ma.lin1.bias, dz1)
    dhn = K.merge_patches(dmerged, n_images, ma.n_h_patches, ma.n_w_patches, ma.m, inverse=True)
    return _ln_bwd(ma.norm, xl, meanm, rsigm, dhn, None)


class _PieceFn(torch.autograd.Function):
    """One autograd node over a (forward, backward) pair of this file: ``fwd(x) -> (y, saved)``, ``bwd(saved, dy) -> dx``."""

    @staticmethod
    def forward(ctx, x, keep, fwd, bwd, *params):
        y, saved = fwd(x)
        ctx.bwd, ctx.saved, ctx.n = bwd, (saved if keep else None), len(params)
        ctx.need_dx = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.saved is None:
            raise RuntimeError("vision module: backward through a forward that ran without grad mode")
        dx = ctx.bwd(ctx.saved, dy)
        ctx.saved = None
        return (dx if ctx.need_dx else None, None, None, None) + (None,) * ctx.n


def _run_piece(mod, x, fwd, bwd):
    L.require_gpu(x)
    if not hasattr(mod, "_param_list"):
        object.__setattr__(mod, "_param_list", list(mod.parameters()))
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mod._param_list))
    return _PieceFn.apply(x, keep, fwd, bwd, *mod._param_list)


# ------------------------------------------------------------------------------------------- modules (reference API)
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

    def forward(self, x):
        """(b, c, t, h, w) pixels -> (b, t/tp * gh * gw, emb_dim) patch embeddings (reference :88-107)."""
        b = x.shape[0]

        def fwd(px):
            y, rows = _patch_fwd(self, px)
            y = y if px.dtype in (F32, torch.uint8) or not px.is_floating_point() else K.cast(y, px.dtype)
            return y.view(b, -1, y.shape[-1]), (rows,)

        def bwd(saved, dy):
            _patch_bwd(self, saved[0], _as_f32_rows(dy))
            return None  # pixels carry no gradient

        return _run_piece(self, x, fwd, bwd)


class Qwen3_5VisionFFN(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.lin1 = nn.Linear(cfg["vision_emb_dim"], cfg["vision_hidden_dim"])
        self.lin2 = nn.Linear(cfg["vision_hidden_dim"], cfg["vision_emb_dim"])
        self.activ = nn.GELU(approximate="tanh")  # kept for module-tree parity; the tanh-GELU runs as a HIP kernel

    def forward(self, x):
        """lin2(gelu_tanh(lin1(x))) on (..., emb) (reference :124-125)."""

        def fwd(t):
            y, saved = _ffn_fwd(self, _as_bf16_rows(t), out_dtype=F32)
            return _like(y, t, y.shape[-1]), saved

        def bwd(saved, dy):
            return _like(_ffn_bwd(self, saved, _as_bf16_rows(dy)), dy, self.lin1.weight.shape[1])

        return _run_piece(self, x, fwd, bwd)


class Qwen3_5VisionAttention(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.d_in = cfg["vision_emb_dim"]
        self.num_heads = cfg["vision_num_heads"]
        self.head_dim = self.d_in // self.num_heads
        self.qkv = nn.Linear(self.d_in, self.d_in * 3, bias=True)
        self.proj = nn.Linear(self.d_in, self.d_in, bias=True)

    def forward(self, x, cos, sin):
        """x (b, seq_len, d_in), cos / sin (seq_len, head_dim) -> (b, seq_len, d_in) (reference :153-198)."""
        b, seq_len, _ = x.shape
        c, s = _rope_tables(cos, sin, seq_len, self.head_dim, x.device)
        tok = _token_rows(seq_len, b, x.device)

        def fwd(t):
            y, saved = _attn_fwd(self, _as_bf16_rows(t), b, seq_len, c, s, tok)
            return _like(y, t, self.d_in), saved

        def bwd(saved, dy):
            return _like(_attn_bwd(self, saved, _as_bf16_rows(dy)), dy, self.d_in)

        return _run_piece(self, x, fwd, bwd)


class Qwen3_5VisionTransformerBlock(nn.Module):
    def __init__(self, cfg):
        super().__init__()
        self.norm1 = nn.LayerNorm(cfg["vision_emb_dim"], eps=1e-6)
        self.norm2 = nn.LayerNorm(cfg["vision_emb_dim"], eps=1e-6)
        self.att = Qwen3_5VisionAttention(cfg)
        self.ffn = Qwen3_5VisionFFN(cfg)

    def forward(self, x, cos, sin):
        """LayerNorm -> attention -> residual -> LayerNorm -> FFN -> residual on (b, seq_len, d_in) (reference :218-241)."""
        b, seq_len, d = x.shape
        c, s = _rope_tables(cos, sin, seq_len, self.att.head_dim, x.device)
        tok = _token_rows(seq_len, b, x.device)

        def fwd(t):
            y, saved = _block_fwd(self, _as_f32_rows(t), b, seq_len, c, s, tok)
            return _like(y, t, d), saved

        def bwd(saved, dy):
            return _like(_block_bwd(self, saved, _as_f32_rows(dy)), dy, d)

        return _run_piece(self, x, fwd, bwd)


class ViTMergeAdapter(nn.Module):
    def __init__(self, vit_d_out, llm_d_in, n_height_patches, n_width_patches, spatial_merge_size=2):
        super().__init__()
        self.m = spatial_merge_size
        self.n_h_patches, self.n_w_patches = n_height_patches, n_width_patches
        self.merged_size = vit_d_out * self.m**2
        self.norm = nn.LayerNorm(vit_d_out, eps=1e-6)
        self.lin1 = nn.Linear(self.merged_size, self.merged_size)
        self.activ = nn.GELU()  # module-tree parity; the erf-GELU runs as a HIP kernel
        self.lin2 = nn.Linear(self.merged_size, llm_d_in)

    def forward(self, x):
        """(b, num_patches, vit_d_out) row-major patches -> (b, num_merged_patches, llm_d_in) (reference :411-431)."""
        b, n_patches, d = x.shape
        per_image = self.n_h_patches * self.n_w_patches
        if n_patches % per_image:
            raise ValueError(f"ViTMergeAdapter: {n_patches} patches are not a multiple of the {per_image} patches of one image")
        n_images = b * (n_patches // per_image)

        def fwd(t):
            y, saved = _merge_fwd(self, _as_f32_rows(t), n_images)
            y = y if y.dtype == t.dtype else K.cast(y, t.dtype)
            return y.view(b, -1, y.shape[-1]), saved

        def bwd(saved, dy):
            return _like(_merge_bwd(self, saved, _as_bf16_rows(dy)), dy, d).view(b, n_patches, d)

        return _run_piece(self, x, fwd, bwd)


# ------------------------------------------------------------------------------------------- whole tower
def _forward(m, pixels):
    pe = m.patch_embed
    if pixels.dim() != 5:
        raise ValueError("Qwen3_5VisionModel expects (b, c, t, h, w) pixels")
    B, T = pixels.shape[0], pixels.shape[2]
    nsp, frames = m.n_spatial_patches, T // pe.temporal_patch_size
    S = frames * nsp
    # learned positional embedding of the patch's spatial index, the same for every frame: fused as the GEMM residual
    posr = m.pos_embed.weight.detach()[:nsp].contiguous().repeat(B * frames, 1)
    x, rows = _patch_fwd(pe, pixels, residual=posr)
    tok_pos = _token_rows(nsp, B * frames, x.device)  # RoPE table row of every token
    saved = []
    for blk in m.blocks:
        x, sv = _block_fwd(blk, x, B, S, m.cos, m.sin, tok_pos)
        saved.append(sv)
    out, msaved = _merge_fwd(m.merge_adapter, x, B * frames)
    return out.view(B, -1, out.shape[-1]), (rows, saved, msaved, (B, frames))


def _backward(m, saved_all, dout):
    rows, saved, msaved, (B, frames) = saved_all
    g = dout.reshape(-1, dout.shape[-1]).contiguous()
    dx = _merge_bwd(m.merge_adapter, msaved, g if g.dtype == BF16 else K.cast(g, BF16))
    for blk, sv in zip(reversed(m.blocks), reversed(saved)):
        dx = _block_bwd(blk, sv, dx)
    # positional embedding (summed over batch and frames), then the patch projection
    nsp, d = m.n_spatial_patches, m.emb_dim
    if m.pos_embed.weight.requires_grad:
        gpos = K.colsum(dx.view(B * frames, nsp * d))
        full = torch.zeros_like(m.pos_embed.weight)
        K.copy2d(gpos.view(nsp, d), full[:nsp])
        _acc(m.pos_embed.weight, full)
    _patch_bwd(m.patch_embed, rows, dx)


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
        cos, sin = VisionRoPE.compute_angles_2d(cfg.get("vision_rope_base", 10_000), self.emb_dim // self.num_heads, self.n_height_patches, self.n_width_patches)
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
