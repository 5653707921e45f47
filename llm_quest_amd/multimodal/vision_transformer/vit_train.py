"""ViT forward+backward on HIP kernels (BASELINE config 2: ViT-Base/16 fwd+bwd in bf16).

Dtype flow = the reference under ``torch.autocast(bf16)``: fp32 master parameters and fp32 residual
stream, bf16 MFMA operands (weights cast once per step, LayerNorm outputs written as bf16), bf16 logits.  The whole
encoder is ONE autograd node: forward keeps what backward needs (28 KB per token per layer), backward runs the dgrad /
wgrad GEMMs, flash-attention backward (non-causal, D=64), exact-GELU and LayerNorm(sigma+eps) backward kernels and writes
every parameter gradient (fp32) straight into ``p.grad``.

Dropout (``drop_rate`` > 0 in train mode; reference vit_model.py:146, vit_attention.py:79, vit_transformer_block.py:117,124) runs on
the HIP path with counter-based Philox masks that are regenerated, never stored (``llm_quest_amd/rng.py``): the embedding site is one
element-wise pass, the two residual sites fuse the residual add (forward) and the bf16 cast of the incoming gradient (backward) into
their pass, and the attention-weight site runs inside the attention kernels (``mi355_attn_dropout_fwd/bwd``).  With ``drop_rate`` 0 or
in eval mode nothing changes: residual adds stay GEMM epilogues and attention stays on the tuned kernels.
"""

import torch

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd import ops, rng
from llm_quest_amd.multimodal.vision_transformer.vit_attention import bf16_cached, f32_cat_cached

BF16, F32 = torch.bfloat16, torch.float32


# ------------------------------------------------------------------------------------------- gradient sinks
def _grad_buf(p):
    """(destination fp32 tensor, accumulate?) for parameter p; attaches a fresh .grad when there is none."""
    if p.grad is None:
        p.grad = torch.empty_like(p)
        return p.grad, False
    return p.grad, True


def _acc(p, g):
    """p.grad (+)= g  (g: contiguous fp32 with p.numel() elements)."""
    if not p.requires_grad:
        return
    dst, acc = _grad_buf(p)
    L.call("mi355_reduce_rows_f32", 1, p.numel(), L.ptr(g), L.ptr(dst), L.DT_F32, int(acc))


def _wgrad(p, dy, x, defer=None):
    """p.grad[N,K] (+)= dy^T x  (TN GEMM, fp32 output).  With ``defer`` (a list) the problem is recorded for one grouped
    launch per block (``ops._flush_wgrads``)."""
    if not p.requires_grad:
        return
    dst, acc = _grad_buf(p)
    view = dst.view(dy.shape[1], x.shape[1])
    if defer is not None and ops.GROUP_WGRADS:
        defer.append((dy, x, view, view if acc else None))
        return
    K.gemm(L.GEMM_TN, dy, x, out=view, residual=view if acc else None)


def _bgrad(p, dy):
    if p is None or not p.requires_grad:
        return
    dst, acc = _grad_buf(p)
    K.colsum(dy, out=dst.view(-1), accumulate=acc)


def _ln_bwd(ln, x2d, mean, rsig, dy, dres):
    dx, dsc, dsh = K.layernorm_bwd(x2d, ln.scale.detach(), mean, rsig, dy, dres=dres, eps=ln.eps)
    _acc(ln.scale, dsc.contiguous())
    _acc(ln.shift, dsh.contiguous())
    return dx



# ------------------------------------------------------------------------------------------- pieces
# Every sub-module of the reference's ViT is an ordinary autograd module (vit_attention.py:8-91, vit_transformer_block.py:12-127,
# vit_model.py:19-89).  Here each is a (forward, backward) pair over HIP kernels; ``ViTModel`` chains the block pair inside ONE autograd
# node (``ViTTrainFn``), and each class's own ``forward`` wraps its pair in one node (``run_piece``), so the pieces train stand-alone too.
def as_f32_rows(x):
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    return x2 if x2.dtype == F32 else K.cast(x2, F32)


def as_bf16_rows(x):
    x2 = x.reshape(-1, x.shape[-1])
    x2 = x2 if x2.is_contiguous() else x2.contiguous()
    return x2 if x2.dtype == BF16 else K.cast(x2, BF16)


def like(y2d, ref, last):
    """rows [n, last] -> ref's dtype and leading dimensions."""
    y = y2d if y2d.dtype == ref.dtype else K.cast(y2d, ref.dtype)
    return y.view(*ref.shape[:-1], last)


def att_forward(att, h1, B, S, training, residual=None, out_dtype=F32, p_res=0.0):
    """LayerNormed tokens bf16 [B*S, d_in] -> out_proj(attention) [B*S, d_out] (+ residual / residual-dropout as the block wires it)."""
    d, H, Dh = att.d_out, att.num_heads, att.head_dim
    wqkv = bf16_cached(att, "wqkv", [att.w_queries.weight, att.w_keys.weight, att.w_values.weight])
    bqkv = f32_cat_cached(att, "bqkv", [att.w_queries.bias, att.w_keys.bias, att.w_values.bias]) if att.w_queries.bias is not None else None
    qkv = K.gemm(L.GEMM_NT, h1, wqkv, bias=bqkv)
    p_att = att.dropout.p if training else 0.0
    s_att = s_proj = None
    if p_att > 0:  # dropout on the softmax weights, inside the kernel
        s_att = rng.draw()
        ctx, lse = K.attn_dropout_fwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], B, S, H, H, Dh, p_att, *s_att, causal=False, scale=att.att_scaling)
    else:
        ctx, lse = K.attn_fwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], B, S, H, H, Dh, key_mask=None, causal=False, scale=att.att_scaling)
    wo = bf16_cached(att, "wo", [att.out_proj.weight])
    if p_res > 0:  # x2 = x + dropout(proj): the residual add moves from the GEMM epilogue into the dropout pass
        s_proj = rng.draw()
        y = K.dropout(K.gemm(L.GEMM_NT, ctx, wo, bias=att.out_proj.bias.detach(), out_dtype=F32), p_res, *s_proj, residual=residual)
    else:
        y = K.gemm(L.GEMM_NT, ctx, wo, bias=att.out_proj.bias.detach(), residual=residual, out_dtype=out_dtype)
    return y, (h1, qkv, ctx, lse, (B, S), (p_att, s_att, p_res, s_proj))


def att_backward(att, saved, dy, wg=None):
    """dy: gradient of the piece's output (fp32 or bf16 rows).  Returns d(h1) bf16; weight / bias gradients land in ``.grad``."""
    h1, qkv, ctx, lse, (B, S), (p_att, s_att, p_res, s_proj) = saved
    d, H, Dh = att.d_out, att.num_heads, att.head_dim
    if p_res > 0:
        dyb = K.dropout(dy if dy.dtype == F32 else K.cast(dy, F32), p_res, *s_proj, out_dtype=BF16)
    else:
        dyb = dy if dy.dtype == BF16 else K.cast(dy, BF16)
    own = wg is None
    wg = [] if own else wg
    dctx = K.dgrad(dyb, bf16_cached(att, "wo", [att.out_proj.weight]))
    _wgrad(att.out_proj.weight, dyb, ctx, wg)
    _bgrad(att.out_proj.bias, dyb)
    dqkv = torch.empty_like(qkv)
    if p_att > 0:
        K.attn_dropout_bwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], ctx, dctx, lse, B, S, H, H, Dh,
                           dqkv[:, :d], dqkv[:, d : 2 * d], dqkv[:, 2 * d :], p_att, *s_att, causal=False, scale=att.att_scaling)
    else:
        K.attn_bwd(qkv[:, :d], qkv[:, d : 2 * d], qkv[:, 2 * d :], ctx, dctx, lse, B, S, H, H, Dh,
                   dqkv[:, :d], dqkv[:, d : 2 * d], dqkv[:, 2 * d :], key_mask=None, causal=False, scale=att.att_scaling)
    wqkv = bf16_cached(att, "wqkv", [att.w_queries.weight, att.w_keys.weight, att.w_values.weight])
    dh1 = K.dgrad(dqkv, wqkv)
    for i, lin in enumerate((att.w_queries, att.w_keys, att.w_values)):
        _wgrad(lin.weight, dqkv[:, i * d : (i + 1) * d], h1, wg)
    if att.w_queries.bias is not None:
        gb = K.colsum(dqkv)
        for i, lin in enumerate((att.w_queries, att.w_keys, att.w_values)):
            _acc(lin.bias, gb[i * d : (i + 1) * d])
    if own:
        ops._flush_wgrads(wg)
    return dh1


def ffn_forward(ffn, h2, residual=None, out_dtype=F32, p_res=0.0):
    """bf16 [M, d] -> lin2(gelu(lin1(h2))) (+ residual / residual-dropout as the block wires it)."""
    w1 = bf16_cached(ffn, "w1", [ffn.layers[0].weight])
    y1, f = K.gemm_gelu_dual(h2, w1, bias=ffn.layers[0].bias.detach())  # Linear + GELU in one launch (pre-activation kept for the backward)
    w2 = bf16_cached(ffn, "w2", [ffn.layers[2].weight])
    s_ffn = None
    if p_res > 0:
        s_ffn = rng.draw()
        y = K.dropout(K.gemm(L.GEMM_NT, f, w2, bias=ffn.layers[2].bias.detach(), out_dtype=F32), p_res, *s_ffn, residual=residual)
    else:
        y = K.gemm(L.GEMM_NT, f, w2, bias=ffn.layers[2].bias.detach(), residual=residual, out_dtype=out_dtype)
    return y, (h2, y1, f, (p_res, s_ffn))


def ffn_backward(ffn, saved, dy, wg=None):
    h2, y1, f, (p_res, s_ffn) = saved
    if p_res > 0:  # the dropout site's backward is the same mask on the gradient, fused with the bf16 cast
        dyb = K.dropout(dy if dy.dtype == F32 else K.cast(dy, F32), p_res, *s_ffn, out_dtype=BF16)
    else:
        dyb = dy if dy.dtype == BF16 else K.cast(dy, BF16)
    own = wg is None
    wg = [] if own else wg
    dy1 = K.gemm_dgrad_gelu_bwd(dyb, bf16_cached(ffn, "w2", [ffn.layers[2].weight]), y1)  # GELU backward in the dgrad epilogue
    _wgrad(ffn.layers[2].weight, dyb, f, wg)
    _bgrad(ffn.layers[2].bias, dyb)
    dh2 = K.dgrad(dy1, bf16_cached(ffn, "w1", [ffn.layers[0].weight]))
    _wgrad(ffn.layers[0].weight, dy1, h2, wg)
    _bgrad(ffn.layers[0].bias, dy1)
    if own:
        ops._flush_wgrads(wg)
    return dh2


def block_forward(blk, x, B, S, training):
    """Pre-LN encoder block on the fp32 residual stream x [B*S, d] (reference vit_transformer_block.py:106-127)."""
    p_res = blk.dropout.p if training else 0.0
    h1, mean1, rsig1 = K.layernorm_fwd(x, blk.ln_1.scale.detach(), blk.ln_1.shift.detach(), out_dtype=BF16, eps=blk.ln_1.eps, want_stats=True)
    x2, sv_att = att_forward(blk.att, h1, B, S, training, residual=x, out_dtype=F32, p_res=p_res)
    h2, mean2, rsig2 = K.layernorm_fwd(x2, blk.ln_2.scale.detach(), blk.ln_2.shift.detach(), out_dtype=BF16, eps=blk.ln_2.eps, want_stats=True)
    x3, sv_ffn = ffn_forward(blk.ffn, h2, residual=x2, out_dtype=F32, p_res=p_res)
    return x3, (x, mean1, rsig1, sv_att, x2, mean2, rsig2, sv_ffn)


def block_backward(blk, saved, dx, B, S):
    """dx fp32 [B*S, d] = gradient of the block's output; returns the gradient of its input; one grouped launch for the six weight gradients."""
    x, mean1, rsig1, sv_att, x2, mean2, rsig2, sv_ffn = saved
    wg = []
    dh2 = ffn_backward(blk.ffn, sv_ffn, dx, wg)
    dx2 = _ln_bwd(blk.ln_2, x2, mean2, rsig2, dh2, dx)
    dh1 = att_backward(blk.att, sv_att, dx2, wg)
    dxin = _ln_bwd(blk.ln_1, x, mean1, rsig1, dh1, dx2)
    ops._flush_wgrads(wg)
    return dxin


class PieceFn(torch.autograd.Function):
    """One autograd node over a (forward, backward) pair: ``fwd(x) -> (y, saved)``, ``bwd(saved, dy) -> dx``."""

    @staticmethod
    def forward(ctx, x, keep, fwd, bwd, *params):
        y, saved = fwd(x)
        ctx.bwd, ctx.saved, ctx.n = bwd, (saved if keep else None), len(params)
        ctx.need_dx = x.requires_grad
        return y

    @staticmethod
    def backward(ctx, dy):
        if ctx.saved is None:
            raise RuntimeError("ViT module: backward called twice, or through a forward that ran without grad mode")
        dx = ctx.bwd(ctx.saved, dy if dy.is_contiguous() else dy.contiguous())
        ctx.saved = None
        return (dx if ctx.need_dx else None, None, None, None) + (None,) * ctx.n


def run_piece(mod, x, fwd, bwd):
    L.require_gpu(x)
    if not hasattr(mod, "_param_list"):
        object.__setattr__(mod, "_param_list", list(mod.parameters()))
    keep = torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in mod._param_list))
    return PieceFn.apply(x, keep, fwd, bwd, *mod._param_list)


# ------------------------------------------------------------------------------------------- forward
def vit_forward_train(m, img, output_hidden_states):
    pe = m.patch_embedding
    B = img.shape[0]
    S, d = pe.num_patches + 1, m.pos_embedding.shape[-1]
    p_emb = m.dropout.p if m.training else 0.0
    rows = K.patchify(img.contiguous().to(F32), pe.patch_size, out_dtype=BF16)
    wconv = bf16_cached(pe, "wconv", [pe.conv_proj.weight])
    proj = K.gemm(L.GEMM_NT, rows, wconv, bias=pe.conv_proj.bias.detach(), out_dtype=F32)
    x = K.vit_embed_assemble(proj, pe.cls_token.detach().reshape(-1).contiguous(), m.pos_embedding.detach().reshape(S, d).contiguous(), B, S, d).view(B * S, d)
    s_emb = None
    if p_emb > 0:
        s_emb = rng.draw()
        x = K.dropout(x, p_emb, *s_emb)
    saved_blocks = []
    for blk in m.transformer_blocks:
        x, sv = block_forward(blk, x, B, S, m.training)
        saved_blocks.append(sv)
    ln = m.final_ln
    if output_hidden_states:
        out, meanf, rsigf = K.layernorm_fwd(x, ln.scale.detach(), ln.shift.detach(), out_dtype=F32, eps=ln.eps, want_stats=True)
        tail = ("hidden", x, meanf, rsigf)
        out = out.view(B, S, d)
    else:
        cls_rows = torch.empty((B, d), dtype=F32, device=x.device)
        K.copy2d(x.view(B, S * d)[:, :d], cls_rows)
        cls_n, meanf, rsigf = K.layernorm_fwd(cls_rows, ln.scale.detach(), ln.shift.detach(), out_dtype=BF16, eps=ln.eps, want_stats=True)
        wc = bf16_cached(m, "wcls", [m.classifier.weight])
        out = K.gemm(L.GEMM_NT, cls_n, wc, bias=m.classifier.bias.detach(), out_dtype=BF16)
        tail = ("logits", cls_rows, meanf, rsigf, cls_n)
    return out, (rows, saved_blocks, tail, (B, S, d), (p_emb, s_emb))


# ------------------------------------------------------------------------------------------- backward
def vit_backward(m, saved, dout):
    rows, saved_blocks, tail, (B, S, d), (p_emb, s_emb) = saved
    pe = m.patch_embedding
    ln = m.final_ln
    if tail[0] == "hidden":
        _, xl, meanf, rsigf = tail
        g = dout.reshape(B * S, d)
        g = g if g.is_contiguous() else g.contiguous()
        dx = _ln_bwd(ln, xl, meanf, rsigf, g if g.dtype in (F32, BF16) else g.to(F32), None)
    else:
        _, cls_rows, meanf, rsigf, cls_n = tail
        dlog = dout.contiguous()
        dlog = dlog if dlog.dtype == BF16 else K.cast(dlog, BF16)
        wc = bf16_cached(m, "wcls", [m.classifier.weight])
        ncls = dlog.shape[1]
        if ncls % 8:  # the dgrad / wgrad GEMMs need 16-byte rows: pad the class dimension with zero columns
            pad = (ncls + 7) // 8 * 8
            dl_p = torch.zeros((B, pad), dtype=BF16, device=dlog.device)
            K.copy2d(dlog, dl_p[:, :ncls])
            wc_p = torch.zeros((pad, d), dtype=BF16, device=dlog.device)
            K.copy2d(wc, wc_p[:ncls])
        else:
            dl_p, wc_p = dlog, wc
        dcls_n = K.dgrad(dl_p, wc_p)
        if m.classifier.weight.requires_grad:
            gw = K.gemm(L.GEMM_TN, dl_p, cls_n, out_dtype=F32)  # [pad, d]
            _acc(m.classifier.weight, gw[:ncls].contiguous())
        _bgrad(m.classifier.bias, dlog)
        dcls = _ln_bwd(ln, cls_rows, meanf, rsigf, dcls_n, None)
        dx = torch.zeros((B * S, d), dtype=F32, device=dcls.device)
        K.copy2d(dcls, dx.view(B, S * d)[:, :d])
    for blk, sv in zip(reversed(m.transformer_blocks), reversed(saved_blocks)):
        dx = block_backward(blk, sv, dx, B, S)
    # ---- embedding: dropout site, pos / cls sums over the batch, patch projection wgrad
    if p_emb > 0:
        dx = K.dropout(dx, p_emb, *s_emb)
    gpos = K.colsum(dx.view(B, S * d))  # sum_b dh[b, s, :]
    _acc(m.pos_embedding, gpos)
    _acc(pe.cls_token, gpos[:d].contiguous())
    npatch = S - 1
    dproj = torch.empty((B, npatch * d), dtype=F32, device=dx.device)
    K.copy2d(dx.view(B, S * d)[:, d:], dproj)
    dproj_b = K.cast(dproj, BF16).view(B * npatch, d)
    _wgrad(pe.conv_proj.weight, dproj_b, rows)
    _bgrad(pe.conv_proj.bias, dproj_b)


class ViTTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, img, model, output_hidden_states, *params):
        out, saved = vit_forward_train(model, img, output_hidden_states)
        ctx.model, ctx.saved = model, saved
        return out

    @staticmethod
    def backward(ctx, dout):
        if ctx.saved is None:
            raise RuntimeError("ViTTrainFn: backward called twice")
        vit_backward(ctx.model, ctx.saved, dout)
        ctx.saved = None
        return (None, None, None) + (None,) * len(ctx.model._param_list)


def needs_training_path(model):
    """Trainable parameters under grad mode, or train-mode dropout (which the forward-only path does not carry)."""
    if torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
        return True
    return model.training and (model.dropout.p > 0 or any(b.dropout.p > 0 or b.att.dropout.p > 0 for b in model.transformer_blocks))


def run_train(model, img, output_hidden_states):
    if not hasattr(model, "_param_list"):
        object.__setattr__(model, "_param_list", list(model.parameters()))
    return ViTTrainFn.apply(img, model, output_hidden_states, *model._param_list)
