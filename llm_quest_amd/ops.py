"""Autograd layer over the HIP kernels.

Design (MI355X-first):
  * activations are token-major 2-D bf16 tensors [B*S, width]; heads are never transposed in HBM;
  * one ``torch.autograd.Function`` per transformer block (not per op): forward enqueues ~10 kernels, backward ~20,
    so the autograd graph has ~30 nodes per step and Python overhead stays far below the GPU time;
  * weight gradients are written by the wgrad GEMMs straight into the block's gradient arena (``p.grad`` is a view of
    it), which is also the RCCL bucket -- the Functions return None for parameters;
  * after a block's backward finishes, ``blk._grad_ready(blk)`` (if set by the data-parallel engine) is called so the
    bucket's all-reduce can start on the communication stream while earlier blocks are still in backward.
"""

import os

import torch

from . import _lib as L
from . import kernels as K
from .arena import ParamArena

BF16, F32 = torch.bfloat16, torch.float32


# ----------------------------------------------------------------------------------------------- arenas
def arena_for(module):
    """Arena that owns ``module``'s parameters; created over the module's own parameters on first use."""
    ar = getattr(module, "_arena", None)
    if ar is None:
        ar = ParamArena(list(module.named_parameters()))
        object.__setattr__(module, "_arena", ar)
        for m in module.modules():
            if m is not module and getattr(m, "_arena", None) is None:
                object.__setattr__(m, "_arena", ar)
    ar.ensure()
    return ar


FUSE_SWIGLU_FWD = os.environ.get("MI355_FUSE_SWIGLU_FWD", "1") != "0"  # 0: separate projection GEMM + swiglu_fwd kernel (A/B measurements)
FUSE_SWIGLU_BWD = os.environ.get("MI355_FUSE_SWIGLU_BWD", "1") != "0"  # 0: separate dgrad GEMM + swiglu_bwd kernels (A/B measurements)
GROUP_WGRADS = os.environ.get("MI355_GROUP_WGRADS", "1") != "0"  # 0: one launch per weight gradient (A/B measurements)


def _wgrad(arena, first, last, dy, x, defer=None):
    """dW[first..last] (+)= dy^T x, written into the arena gradient (TN GEMM).  With `defer` (a list) the problem is only
    recorded: the caller flushes the list with ONE grouped launch (``_flush_wgrads``) once the block's backward is done."""
    if not first.requires_grad:
        return
    view, acc = arena.grad_target(first, last)
    if defer is not None and GROUP_WGRADS:
        defer.append((dy, x, view, view if acc else None))
        return
    K.gemm(L.GEMM_TN, dy, x, out=view, residual=view if acc else None)


def _flush_wgrads(defer):
    """The weight gradients of one block, each too small to fill the chip, as one grouped TN launch per output dtype."""
    by_dtype = {}
    for q in defer:
        by_dtype.setdefault(q[2].dtype, []).append(q)
    for qs in by_dtype.values():
        K.gemm_grouped(L.GEMM_TN, qs)
    defer.clear()


def _vecgrad(arena, p):
    if not p.requires_grad:
        return None, False
    return arena.grad_target(p)


class Runtime:
    """Per-forward constants shared by all blocks: positions, key mask, RoPE tables, shape."""

    __slots__ = ("B", "S", "pos", "key_mask", "cos", "sin")

    def __init__(self, B, S, pos, key_mask, cos, sin):
        self.B, self.S, self.pos, self.key_mask, self.cos, self.sin = B, S, pos, key_mask, cos, sin


_pos_cache = {}


def make_runtime(B, S, device, cos, sin, attn_mask=None, position_ids=None):
    if position_ids is not None:
        if tuple(position_ids.shape) != (B, S):
            raise ValueError(f"position_ids must be (b, s) = {(B, S)}, got {tuple(position_ids.shape)}")
        pos = position_ids.to(device=device, dtype=torch.int32).reshape(-1).contiguous()
    else:
        key = (B, S, str(device))
        pos = _pos_cache.get(key)
        if pos is None:
            pos = torch.arange(S, dtype=torch.int32, device=device).repeat(B)
            _pos_cache[key] = pos
    if S > cos.shape[0]:
        raise ValueError(f"sequence length {S} exceeds context_length {cos.shape[0]}")
    km = None
    if attn_mask is not None:
        if tuple(attn_mask.shape) != (B, S):
            raise ValueError(f"attn_mask must be (b, s) = {(B, S)}, got {tuple(attn_mask.shape)}")
        km = attn_mask.to(device=device, dtype=torch.uint8).contiguous()
    return Runtime(B, S, pos, km, cos, sin)


# ----------------------------------------------------------------------------------------------- attention half
def attention_forward(att, arena, h1, rt):
    """GroupedQueryAttention up to (not including) out_proj.  h1 [M,d] -> ctx [M,Hq*D] + what backward needs."""
    Hq, Hkv, D = att.num_heads, att.num_kv_groups, att.head_dim
    wqkv = arena.fused(att.w_queries.weight, att.w_values.weight)
    qkv = K.gemm(L.GEMM_NT, h1, wqkv)
    q, k, rstd = K.qknorm_rope_fwd(qkv, att.q_norm.weight, att.k_norm.weight, rt.cos, rt.sin, rt.pos, Hq, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D :]
    ctx, lse = K.attn_fwd(q, k, v, rt.B, rt.S, Hq, Hkv, D, key_mask=rt.key_mask, causal=True, scale=att.att_scaling)
    return ctx, (qkv, q, k, rstd, lse)


def attention_backward(att, arena, h1, ctx, saved, dctx, rt, defer=None, delta=None):
    """Returns dh1 [M,d]; writes the QKV / QK-norm weight gradients.  ``delta``: the softmax backward's row sums when the out-projection's
    dgrad already left them (kernels.dgrad_attn_delta)."""
    Hq, Hkv, D = att.num_heads, att.num_kv_groups, att.head_dim
    qkv, q, k, rstd, lse = saved
    v = qkv[:, (Hq + Hkv) * D :]
    dqkv = torch.empty_like(qkv)
    dk = torch.empty_like(k)
    # the dQ pass ends in the QK-norm + RoPE backward of the query heads when it can (head_dim 128, scratch for the one-product form) ...
    dqw = K.attn_bwd_qnorm(q, k, v, ctx, dctx, lse, rt.B, rt.S, Hq, Hkv, D, dk, dqkv[:, (Hq + Hkv) * D :], qkv, att.q_norm.weight, rt.cos, rt.sin, rt.pos, rstd, dqkv,
                           key_mask=rt.key_mask, causal=True, scale=att.att_scaling, delta=delta)
    if dqw is not None:
        _, dkw = K.qknorm_rope_bwd(qkv, att.q_norm.weight, att.k_norm.weight, rt.cos, rt.sin, rt.pos, rstd, None, dk, dqkv, Hq, Hkv, D)
    else:  # ... otherwise dQ is a matrix and one kernel handles every head
        dq = torch.empty_like(q)
        K.attn_bwd(q, k, v, ctx, dctx, lse, rt.B, rt.S, Hq, Hkv, D, dq, dk, dqkv[:, (Hq + Hkv) * D :], key_mask=rt.key_mask, causal=True, scale=att.att_scaling)
        dqw, dkw = K.qknorm_rope_bwd(qkv, att.q_norm.weight, att.k_norm.weight, rt.cos, rt.sin, rt.pos, rstd, dq, dk, dqkv, Hq, Hkv, D)
    for p, g in ((att.q_norm.weight, dqw), (att.k_norm.weight, dkw)):
        view, acc = _vecgrad(arena, p)
        if view is not None:
            K.add_f32_to_bf16(g.contiguous(), view if acc else None, view)
    wqkv = arena.fused(att.w_queries.weight, att.w_values.weight)
    dh1 = K.dgrad(dqkv, wqkv)
    _wgrad(arena, att.w_queries.weight, att.w_values.weight, dqkv, h1, defer)
    return dh1


# ----------------------------------------------------------------------------------------------- Qwen3 block
def _take_rows(x2d, B, S, rows):
    """Rows [lo, hi) of every sample of a token-major [B*S, d] matrix as a contiguous [B*(hi-lo), d] one (one strided device copy)."""
    lo, hi = rows
    d = x2d.shape[1]
    out = torch.empty((B * (hi - lo), d), dtype=x2d.dtype, device=x2d.device)
    K.copy2d(x2d.view(B, S * d)[:, lo * d : hi * d], out.view(B, (hi - lo) * d))
    return out


def _put_rows(g2d, B, S, rows):
    """The adjoint: a zero [B*S, d] matrix with ``g2d`` in rows [lo, hi) of every sample."""
    lo, hi = rows
    d = g2d.shape[1]
    full = torch.zeros((B * S, d), dtype=g2d.dtype, device=g2d.device)
    K.copy2d(g2d.view(B, (hi - lo) * d), full.view(B, S * d)[:, lo * d : hi * d])
    return full


def block_forward(blk, x, rt, keep, rows=None):
    """``rows`` = (lo, hi): only these rows of every sample are needed downstream (the decoder's LAST block in the early-fusion step: the loss reads the
    512 rows that predict text tokens, so the 197 others feed nothing once their keys and values have been used).  The attention half runs on the
    whole sequence, the FFN half -- 60 % of the block's arithmetic -- on the kept rows only, and the block returns [B*(hi-lo), d].  Same results on
    the kept rows, bit for bit: every kernel of the FFN half works row by row."""
    arena = arena_for(blk)
    att, ffn = blk.att, blk.ffn
    F_ = ffn.lin1.weight.shape[0]
    h1, rstd1 = K.rmsnorm_fwd(x, blk.norm1.weight)
    ctx, att_saved = attention_forward(att, arena, h1, rt)
    x2 = K.gemm(L.GEMM_NT, ctx, att.out_proj.weight, residual=x)
    if rows is not None:
        x2 = _take_rows(x2, rt.B, rt.S, rows)
    h2, rstd2 = K.rmsnorm_fwd(x2, blk.norm2.weight)
    if FUSE_SWIGLU_FWD and F_ % 32 == 0:  # the activation is the projection's epilogue (gu is still written: the backward needs it)
        gu, a = K.gemm_gateup_swiglu(h2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
    else:
        gu = K.gemm(L.GEMM_NT, h2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
        a = K.swiglu_fwd(gu, F_)
    x3 = K.gemm(L.GEMM_NT, a, ffn.lin2.weight, residual=x2)
    saved = (x, h1, rstd1, ctx, att_saved, x2, h2, rstd2, gu, a) if keep else None
    return x3, saved


def block_backward(blk, saved, dx3, rt, rows=None):
    arena = arena_for(blk)
    att, ffn = blk.att, blk.ffn
    F_ = ffn.lin1.weight.shape[0]
    x, h1, rstd1, ctx, att_saved, x2, h2, rstd2, gu, a = saved
    wg = []  # the four weight-gradient GEMMs run as one grouped launch at the end (their operands stay alive until then)
    # ---- FFN half
    if FUSE_SWIGLU_BWD:  # d(act) never leaves the accumulators: the activation's backward is the dgrad GEMM's epilogue
        dgu = K.gemm_dgrad_swiglu_bwd(dx3, ffn.lin2.weight, gu)
    else:
        dgu = K.swiglu_bwd(gu, K.dgrad(dx3, ffn.lin2.weight), F_)
    _wgrad(arena, ffn.lin2.weight, None, dx3, a, wg)
    wgu = arena.fused(ffn.lin1.weight, ffn.lin_gate.weight)
    dh2 = K.dgrad(dgu, wgu)
    _wgrad(arena, ffn.lin1.weight, ffn.lin_gate.weight, dgu, h2, wg)
    gview, gacc = _vecgrad(arena, blk.norm2.weight)
    dx2, _ = K.rmsnorm_bwd(x2, blk.norm2.weight, rstd2, dh2, dres=dx3, dw_out=gview, dw_accumulate=gacc)
    if rows is not None:  # the FFN half ran on the kept rows: the others receive no gradient from it
        dx2 = _put_rows(dx2, rt.B, rt.S, rows)
    # ---- attention half
    # the softmax backward's row sums (delta) are the epilogue of the out-projection's dgrad when the shape allows
    fused = K.dgrad_attn_delta(dx2, att.out_proj.weight, ctx, att_saved[4], rt.B, rt.S, att.num_heads, att.head_dim)
    dctx, delta = fused if fused is not None else (K.dgrad(dx2, att.out_proj.weight), None)
    _wgrad(arena, att.out_proj.weight, None, dx2, ctx, wg)
    dh1 = attention_backward(att, arena, h1, ctx, att_saved, dctx, rt, wg, delta)
    gview, gacc = _vecgrad(arena, blk.norm1.weight)
    dx, _ = K.rmsnorm_bwd(x, blk.norm1.weight, rstd1, dh1, dres=dx2, dw_out=gview, dw_accumulate=gacc)
    _flush_wgrads(wg)
    hook = getattr(blk, "_grad_ready", None)
    if hook is not None:
        hook(blk)
    return dx


class Qwen3BlockFn(torch.autograd.Function):
    """``keep``: True = keep the block's activations for the backward; "recompute" = gradient checkpointing (the reference's
    ``torch.utils.checkpoint`` around every block, qwen3_model.py:72-80): only the block INPUT stays alive between forward and
    backward, and the backward first re-runs the forward kernels (bit-identical: the kernels are deterministic) to rebuild what
    it needs; False = inference."""

    @staticmethod
    def forward(ctx, x, blk, rt, keep, rows, *params):
        B, S, d = x.shape
        x2 = x.reshape(B * S, d)
        y, saved = block_forward(blk, x2, rt, keep is True, rows)
        ctx.blk, ctx.rt, ctx.saved, ctx.shape, ctx.rows = blk, rt, saved, (B, S, d), rows
        ctx.x_in = x2 if keep == "recompute" else None
        return y.view(B, S if rows is None else rows[1] - rows[0], d)

    @staticmethod
    def backward(ctx, dy):
        B, S, d = ctx.shape
        saved = ctx.saved
        if saved is None and ctx.x_in is not None:
            _, saved = block_forward(ctx.blk, ctx.x_in, ctx.rt, True, ctx.rows)
            ctx.x_in = None
        if saved is None:
            raise RuntimeError("Qwen3BlockFn: backward through a forward that ran without grad mode")
        dy2 = dy.reshape(-1, d)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dx = block_backward(ctx.blk, saved, dy2, ctx.rt, ctx.rows)
        ctx.saved = None
        return (dx.view(B, S, d), None, None, None, None) + (None,) * len(ctx.blk._param_list)


def run_block(blk, x, rt, recompute=False, rows=None):
    if not hasattr(blk, "_param_list"):
        object.__setattr__(blk, "_param_list", list(blk.parameters()))
    L.require_gpu(x)
    if x.dtype != BF16:
        raise TypeError(f"Qwen3 block expects bf16 activations, got {x.dtype}")
    keep = torch.is_grad_enabled()
    if keep and recompute:
        keep = "recompute"
    if rows is not None and not (0 <= rows[0] < rows[1] <= x.shape[1]):
        raise ValueError(f"run_block: rows {rows} outside the sequence of length {x.shape[1]}")
    return Qwen3BlockFn.apply(x, blk, rt, keep, rows, *blk._param_list)


# ----------------------------------------------------------------------------------------------- generic pieces
class RMSNormFn(torch.autograd.Function):
    """PytorchRMSNorm on [..., width] bf16 (standalone use: final_norm, or modules called on their own)."""

    @staticmethod
    def forward(ctx, x, mod, weight):
        arena_for(mod)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        y, rstd = K.rmsnorm_fwd(x2, mod.weight, eps=mod.eps)
        ctx.mod, ctx.saved, ctx.shp = mod, (x2, rstd), shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        mod = ctx.mod
        x2, rstd = ctx.saved
        dy2 = dy.reshape(x2.shape)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        gview, gacc = _vecgrad(arena_for(mod), mod.weight)
        dx, _ = K.rmsnorm_bwd(x2, mod.weight, rstd, dy2, dw_out=gview, dw_accumulate=gacc)
        return dx.view(ctx.shp), None, None


class LinearFn(torch.autograd.Function):
    """y = x W^T (+ bias) on [..., K] bf16; weight gradient goes to the owner's arena."""

    @staticmethod
    def forward(ctx, x, owner, weight, bias_f32, gelu):
        arena_for(owner)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        ctx.pre = None
        if gelu and (ctx.needs_input_grad[0] or ctx.needs_input_grad[2]):
            ctx.pre, y = K.gemm_gelu_dual(x2, weight, bias=bias_f32)  # the pre-activation stays for the backward (one launch writes both)
        else:
            y = K.gemm(L.GEMM_NT, x2, weight, bias=bias_f32, gelu=gelu)
        ctx.owner, ctx.weight, ctx.x2, ctx.shp, ctx.gelu = owner, weight, x2, shp, gelu
        return y.view(*shp[:-1], weight.shape[0])

    @staticmethod
    def backward(ctx, dy):
        w = ctx.weight
        dy2 = dy.reshape(-1, w.shape[0])
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        if ctx.gelu:
            dy2 = K.gelu_bwd(ctx.pre, dy2)
        dx = K.dgrad(dy2, w)
        _wgrad(arena_for(ctx.owner), w, None, dy2, ctx.x2)
        return dx.view(ctx.shp), None, None, None, None


class EmbeddingFn(torch.autograd.Function):
    """emb_dict gather (bit-exact) with a scatter-add backward into the (possibly tied) weight gradient."""

    @staticmethod
    def forward(ctx, ids, owner, weight):
        arena_for(owner)
        out = K.embedding_fwd(ids, weight)
        ctx.owner, ctx.weight, ctx.ids = owner, weight, ids
        return out.view(*ids.shape, weight.shape[1])

    @staticmethod
    def backward(ctx, dy):
        w = ctx.weight
        if w.requires_grad:
            dy2 = dy.reshape(-1, w.shape[1])
            dy2 = dy2 if dy2.stride(1) == 1 else dy2.contiguous()
            arena = arena_for(ctx.owner)
            view, accum = arena.grad_target(w)
            if w.dtype == BF16 and dy2.dtype == BF16 and w.shape[1] % 8 == 0:
                # deterministic: sorted ids, one wave per run of equal ids, no atomics, only the touched rows move.  Under data parallelism with a
                # split head / embedding bucket (ddp.GradSync.early_tail) every rank sums ALL ranks' token rows, scaled by 1 / world.
                from . import ddp

                ids, rows, scale = ctx.ids, dy2, 1.0
                sync = ddp.active()
                if sync is not None and sync.splits(arena):
                    ids, rows, scale = sync.gather_embedding(ctx.ids, dy2)
                if not accum:
                    view.zero_()
                K.embedding_bwd_sorted(ids, rows, view, True, scale)
            else:  # fp32 tables: the dense accumulator with fp32 atomics
                acc = torch.zeros(w.shape, dtype=F32, device=w.device)
                K.embedding_bwd(ctx.ids, dy2, acc)
                K.add_f32_to_bf16(acc, view if accum else None, view)
        return None, None, None


class CrossEntropyFn(torch.autograd.Function):
    """F.cross_entropy(logits2d, targets, ignore_index=-100) on bf16 logits; loss returned in fp32."""

    @staticmethod
    def forward(ctx, logits, targets):
        loss_rows, _ = K.cross_entropy(logits, targets, want_grad=False)
        out3 = K.ce_finalize(loss_rows, targets)
        ctx.saved = (logits, targets, out3)
        return out3[0].clone()

    @staticmethod
    def backward(ctx, g):
        logits, targets, out3 = ctx.saved
        scale = (out3[2] * g.to(F32)).reshape(1)
        _, dl = K.cross_entropy(logits, targets, want_grad=True, grad_scale=scale, inplace=False)
        return dl, None


class LMHeadLossFn(torch.autograd.Function):
    """Tied LM head + cross entropy on selected hidden rows:  loss = CE(h W^T, targets).

    Forward materialises bf16 logits once (as the reference does), takes the row log-sum-exp, and -- in training --
    overwrites them IN PLACE with d(loss)/d(logits) so the backward is just the two GEMMs.
    """

    @staticmethod
    def forward(ctx, h, targets, owner, weight, keep):
        arena_for(owner)
        logits = K.gemm(L.GEMM_NT, h, weight)
        if keep:
            probe = torch.zeros(h.shape[0], dtype=F32, device=h.device)
            inv_count = K.ce_finalize(probe, targets)[2:3]  # 1/#(targets != -100), stays on the device
            loss_rows, dl = K.cross_entropy(logits, targets, want_grad=True, grad_scale=inv_count, inplace=True)
        else:
            loss_rows, dl = K.cross_entropy(logits, targets, want_grad=False)
        out3 = K.ce_finalize(loss_rows, targets)
        ctx.owner, ctx.weight, ctx.saved = owner, weight, (h, dl)
        return out3[0].clone()

    @staticmethod
    def backward(ctx, g):
        h, dl = ctx.saved
        if dl is None:
            raise RuntimeError("LMHeadLossFn: backward through a forward that ran without grad mode")
        w = ctx.weight
        gs = g.to(F32).reshape(1)
        dh = K.dgrad(dl, w)
        dh = K.scale_bf16(dh, gs, out=dh)
        if w.requires_grad:
            view, acc = arena_for(ctx.owner).grad_target(w)
            K.gemm(L.GEMM_TN, dl, K.scale_bf16(h, gs), out=view, residual=view if acc else None)
        ctx.saved = None
        return dh, None, None, None, None
