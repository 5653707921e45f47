"""Autograd layer of the Qwen3.5 hybrid text stack (BASELINE config 5) over the HIP kernels.

Same design as ``ops.py``: token-major 2-D bf16 activations, ONE autograd node per transformer block, every linear layer of a
half-block fused into a single projection GEMM whose column blocks the row kernels address in place (no splits, no head
transposes in HBM), weight gradients written into the block's arena.

  FusedGatedDeltaNet half:  [w_qkv | w_gate | w_beta | w_alpha] (8224 columns for Qwen3.5-0.8B) is one NT GEMM; the causal
      conv + SiLU, the q/k l2 norms, the gate activations, the delta-rule recurrence and the gated RMSNorm read their columns
      of it by pointer + row pitch, and the backward kernels write their input gradients into the matching columns of ONE
      gradient buffer that then feeds a single dgrad and a single wgrad GEMM.
  MRoPEGatedAttention half: [w_queries_gate | w_keys | w_values] likewise; the q heads sit interleaved with their gates
      (head stride 2*d_h) and are normalised / rotated from there.

The fp32 parameters of the block (``log_A``, ``post_norm.weight``) cannot live in the bf16 arena; their gradients are ordinary
``.grad`` tensors produced by the kernels' reduction passes.
"""

import torch

from . import _lib as L
from . import kernels as K
from . import kernels_q35 as Q
from .arena import ParamArena
from .ops import FUSE_SWIGLU_BWD, FUSE_SWIGLU_FWD, _flush_wgrads, _vecgrad, _wgrad

BF16, F32 = torch.bfloat16, torch.float32


def arena_for_bf16(module):
    """Arena over the module's bf16 parameters (created on first use); fp32 parameters stay stand-alone."""
    ar = getattr(module, "_arena", None)
    if ar is None:
        ar = ParamArena([(n, p) for n, p in module.named_parameters() if p.dtype == BF16])
        for m in module.modules():
            if getattr(m, "_arena", None) is None:
                object.__setattr__(m, "_arena", ar)
    ar.ensure()
    return ar


def _bf16_vec_grad(arena, p, g_f32):
    """p.grad (bf16, in the arena) (+)= g (fp32)."""
    view, acc = _vecgrad(arena, p)
    if view is not None:
        K.add_f32_to_bf16(g_f32.contiguous(), view if acc else None, view)


def _f32_param_grad(p, g_f32):
    """Stand-alone fp32 parameter: p.grad (+)= g."""
    if not p.requires_grad:
        return
    g = g_f32.reshape(p.shape)
    if p.grad is None:
        p.grad = g.clone()
    else:
        Q._reduce_parts(g.reshape(1, -1).contiguous(), out=p.grad.view(-1), accumulate=True)


class Runtime:
    """Per-forward constants shared by all blocks: shape, padding mask, rotary tables and the row of them each token uses."""

    __slots__ = ("B", "S", "key_mask", "cos_t", "sin_t", "pos")

    def __init__(self, B, S, key_mask, cos_t, sin_t, pos):
        self.B, self.S, self.key_mask, self.cos_t, self.sin_t, self.pos = B, S, key_mask, cos_t, sin_t, pos


_pos_cache = {}


def _arange_pos(B, S, device, per_token):
    key = (B, S, str(device), per_token)
    pos = _pos_cache.get(key)
    if pos is None:
        pos = torch.arange(B * S, dtype=torch.int32, device=device) if per_token else torch.arange(S, dtype=torch.int32, device=device).repeat(B)
        _pos_cache[key] = pos
    return pos


def make_runtime(B, S, device, cos, sin, attn_mask=None, position_ids=None, mrope_section=None):
    """position_ids (3, b, s): interleaved MRoPE rows per token (RoPE.apply_mrope); None: rows 0..S-1 of the 1-D table."""
    if S > cos.shape[0]:
        raise ValueError(f"sequence length {S} exceeds context_length {cos.shape[0]}")
    km = None
    if attn_mask is not None:
        if tuple(attn_mask.shape) != (B, S):
            raise ValueError(f"attn_mask must be (b, s) = {(B, S)}, got {tuple(attn_mask.shape)}")
        km = attn_mask.to(device=device, dtype=torch.uint8).contiguous()
    if position_ids is not None:
        if tuple(position_ids.shape) != (3, B, S):
            raise ValueError(f"position_ids must be (3, b, s) = {(3, B, S)}, got {tuple(position_ids.shape)}")
        cos_t, sin_t = Q.mrope_table(cos, sin, position_ids.to(device), mrope_section)
        return Runtime(B, S, km, cos_t, sin_t, _arange_pos(B, S, device, True))
    return Runtime(B, S, km, cos, sin, _arange_pos(B, S, device, False))


# ----------------------------------------------------------------------------------------------- gated attention half
def _att_dims(att):
    H, G, D = att.num_heads, att.num_kv_groups, att.head_dim
    return H, G, D, H * 2 * D, G * D


def gated_attention_forward(att, arena, h1, rt):
    """(MRoPE)GatedAttention up to (not including) out_proj: h1 [M, d] -> gated context [M, H*D] + what backward needs."""
    H, G, D, QG, KV = _att_dims(att)
    proj = K.gemm(L.GEMM_NT, h1, arena.fused(att.w_queries_gate.weight, att.w_values.weight))
    wq, wk = Q.zc_weight(att.q_norm.scale), Q.zc_weight(att.k_norm.scale)
    q, rstd_q = Q.headnorm_rope_fwd(proj[:, :QG], H, D, 2 * D, wq, rt.cos_t, rt.sin_t, rt.pos, eps=att.q_norm.eps)
    k, rstd_k = Q.headnorm_rope_fwd(proj[:, QG : QG + KV], G, D, D, wk, rt.cos_t, rt.sin_t, rt.pos, eps=att.k_norm.eps)
    drop = None
    p_drop = float(getattr(att, "p_dropout", 0.0))
    if p_drop > 0.0:  # SDPA's dropout_p (qwen3_next_attention.py:245-253): Philox masks inside the kernels, regenerated in the backward
        from . import rng

        drop = (p_drop,) + rng.draw()
        if rt.key_mask is not None:  # padded batch: the quirk-mask kernels with the dropout arguments set
            ctx, lse = K.attn_generic_dropout_fwd(q, k, proj[:, QG + KV :], rt.B, rt.S, H, G, D, *drop, key_mask=rt.key_mask)
        else:
            ctx, lse = K.attn_dropout_fwd(q, k, proj[:, QG + KV :], rt.B, rt.S, H, G, D, *drop, causal=True)
    else:
        ctx, lse = Q.attn_generic_fwd(q, k, proj[:, QG + KV :], rt.B, rt.S, H, G, D, key_mask=rt.key_mask)
    gated = Q.sigmoid_gate_fwd(ctx, proj[:, D:QG], H, D, 2 * D)
    return gated, (proj, wq, wk, q, k, rstd_q, rstd_k, ctx, lse, drop)


def gated_attention_backward(att, arena, h1, saved, dgated, rt, defer=None):
    H, G, D, QG, KV = _att_dims(att)
    proj, wq, wk, q, k, rstd_q, rstd_k, ctx, lse, drop = saved
    dproj = torch.empty_like(proj)
    dctx = Q.sigmoid_gate_bwd(ctx, proj[:, D:QG], H, D, 2 * D, dgated, dproj[:, D:QG], 2 * D)
    dq, dk = torch.empty_like(q), torch.empty_like(k)
    if drop is not None and rt.key_mask is not None:
        K.attn_generic_dropout_bwd(q, k, proj[:, QG + KV :], ctx, dctx, lse, rt.B, rt.S, H, G, D, dq, dk, dproj[:, QG + KV :], *drop, key_mask=rt.key_mask)
    elif drop is not None:
        K.attn_dropout_bwd(q, k, proj[:, QG + KV :], ctx, dctx, lse, rt.B, rt.S, H, G, D, dq, dk, dproj[:, QG + KV :], *drop, causal=True)
    else:
        Q.attn_generic_bwd(q, k, proj[:, QG + KV :], ctx, dctx, lse, rt.B, rt.S, H, G, D, dq, dk, dproj[:, QG + KV :], key_mask=rt.key_mask)
    dwq = Q.headnorm_rope_bwd(proj[:, :QG], H, D, 2 * D, wq, rt.cos_t, rt.sin_t, rt.pos, rstd_q, dq, dproj[:, :QG], 2 * D)
    dwk = Q.headnorm_rope_bwd(proj[:, QG : QG + KV], G, D, D, wk, rt.cos_t, rt.sin_t, rt.pos, rstd_k, dk, dproj[:, QG : QG + KV], D)
    _bf16_vec_grad(arena, att.q_norm.scale, dwq)
    _bf16_vec_grad(arena, att.k_norm.scale, dwk)
    w = arena.fused(att.w_queries_gate.weight, att.w_values.weight)
    dh1 = K.dgrad(dproj, w)
    _wgrad(arena, att.w_queries_gate.weight, att.w_values.weight, dproj, h1, defer)
    return dh1


# ----------------------------------------------------------------------------------------------- gated delta net half
def _gdn_dims(att):
    Hqk, Hv, Dk, Dv = att.num_qk_heads, att.num_v_heads, att.qk_head_dim, att.vg_head_dim
    QK, VG = Hqk * Dk, Hv * Dv
    return Hqk, Hv, Dk, Dv, QK, VG, 2 * QK + VG


def gdn_forward(att, arena, h1, rt, keep=True):
    """FusedGatedDeltaNet up to (not including) out_proj: h1 [M, d] -> gated, normed context [M, Hv*Dv]."""
    Hqk, Hv, Dk, Dv, QK, VG, C = _gdn_dims(att)
    if rt.key_mask is not None:  # `x *= attn_mask` (qwen3_5_text_model.py:109-110); h1 is this block's own temporary
        h1 = Q.rowmask(h1, rt.key_mask.view(-1))
    proj = K.gemm(L.GEMM_NT, h1, arena.fused(att.w_qkv.weight, att.w_alpha.weight))
    y = Q.causal_conv_silu_fwd(proj[:, :C], att.conv1d.weight, rt.B, rt.S)
    qn = Q.l2norm_fwd(y[:, :QK], Hqk, Dk)
    kn = Q.l2norm_fwd(y[:, QK : 2 * QK], Hqk, Dk)
    beta, alpha = Q.gdn_gates_fwd(proj[:, C + VG : C + VG + Hv], proj[:, C + VG + Hv :], att.log_A, att.dt_bias)
    o, ck, _ = Q.gated_delta_rule_fwd(qn, kn, y[:, 2 * QK :], beta, alpha, rt.B, rt.S, Hqk, Hv, Dk, Dv, keep=keep)
    gated, rstd_p = Q.gated_rmsnorm_fwd(o, att.post_norm.weight, proj[:, C : C + VG], Hv, Dv, eps=att.post_norm.eps)
    return gated, (h1, proj, y, qn, kn, beta, alpha, o, ck, rstd_p)


def gdn_backward(att, arena, saved, dgated, rt, defer=None):
    Hqk, Hv, Dk, Dv, QK, VG, C = _gdn_dims(att)
    h1, proj, y, qn, kn, beta, alpha, o, ck, rstd_p = saved
    dproj = torch.empty_like(proj)
    d_o, dpw = Q.gated_rmsnorm_bwd(o, att.post_norm.weight, proj[:, C : C + VG], rstd_p, dgated, dproj[:, C : C + VG], Hv, Dv)
    _f32_param_grad(att.post_norm.weight, dpw)
    dy = torch.empty_like(y)
    dqn, dkn, dbeta, dalpha = Q.gated_delta_rule_bwd(qn, kn, y[:, 2 * QK :], beta, alpha, ck, d_o, dy[:, 2 * QK :], rt.B, rt.S, Hqk, Hv, Dk, Dv)
    Q.l2norm_bwd(y[:, :QK], dqn, dy[:, :QK], Hqk, Dk)
    Q.l2norm_bwd(y[:, QK : 2 * QK], dkn, dy[:, QK : 2 * QK], Hqk, Dk)
    dcw = Q.causal_conv_silu_bwd(proj[:, :C], att.conv1d.weight, dy, dproj[:, :C], rt.B, rt.S)
    _bf16_vec_grad(arena, att.conv1d.weight, dcw)
    dlog_A, ddtb = Q.gdn_gates_bwd(proj[:, C + VG : C + VG + Hv], proj[:, C + VG + Hv :], att.log_A, att.dt_bias, dbeta, dalpha,
                                   dproj[:, C + VG : C + VG + Hv], dproj[:, C + VG + Hv :])
    _f32_param_grad(att.log_A, dlog_A)
    _bf16_vec_grad(arena, att.dt_bias, ddtb)
    w = arena.fused(att.w_qkv.weight, att.w_alpha.weight)
    dh1 = K.dgrad(dproj, w)
    _wgrad(arena, att.w_qkv.weight, att.w_alpha.weight, dproj, h1, defer)
    if rt.key_mask is not None:
        dh1 = Q.rowmask(dh1, rt.key_mask.view(-1))
    return dh1


# ----------------------------------------------------------------------------------------------- block
def block_forward(blk, x, rt, keep):
    arena = arena_for_bf16(blk)
    att, ffn = blk.att, blk.ffn
    F_ = ffn.lin1.weight.shape[0]
    w1 = Q.zc_weight(blk.norm1.scale)
    h1, rstd1 = K.rmsnorm_fwd(x, w1, eps=blk.norm1.eps)
    if blk.is_linear:
        mix, att_saved = gdn_forward(att, arena, h1, rt, keep)
    else:
        mix, att_saved = gated_attention_forward(att, arena, h1, rt)
    x2 = K.gemm(L.GEMM_NT, mix, att.out_proj.weight, residual=x)
    w2 = Q.zc_weight(blk.norm2.scale)
    h2, rstd2 = K.rmsnorm_fwd(x2, w2, eps=blk.norm2.eps)
    if FUSE_SWIGLU_FWD and F_ % 32 == 0:  # the activation is the projection's epilogue (gu is still written: the backward needs it)
        gu, a = K.gemm_gateup_swiglu(h2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
    else:
        gu = K.gemm(L.GEMM_NT, h2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
        a = K.swiglu_fwd(gu, F_)
    x3 = K.gemm(L.GEMM_NT, a, ffn.lin2.weight, residual=x2)
    saved = (x, w1, None if blk.is_linear else h1, rstd1, mix, att_saved, x2, w2, h2, rstd2, gu, a) if keep else None  # the GDN half keeps its own (masked) h1
    return x3, saved


def block_backward(blk, saved, dx3, rt):
    arena = arena_for_bf16(blk)
    att, ffn = blk.att, blk.ffn
    F_ = ffn.lin1.weight.shape[0]
    x, w1, h1, rstd1, mix, att_saved, x2, w2, h2, rstd2, gu, a = saved
    wg = []
    # ---- FFN half
    if FUSE_SWIGLU_BWD:  # d(act) never leaves the accumulators: the activation's backward is the dgrad GEMM's epilogue
        dgu = K.gemm_dgrad_swiglu_bwd(dx3, ffn.lin2.weight, gu)
    else:
        dgu = K.swiglu_bwd(gu, K.dgrad(dx3, ffn.lin2.weight), F_)
    _wgrad(arena, ffn.lin2.weight, None, dx3, a, wg)
    dh2 = K.dgrad(dgu, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
    _wgrad(arena, ffn.lin1.weight, ffn.lin_gate.weight, dgu, h2, wg)
    gview, gacc = _vecgrad(arena, blk.norm2.scale)
    dx2, _ = K.rmsnorm_bwd(x2, w2, rstd2, dh2, dres=dx3, dw_out=gview, dw_accumulate=gacc)
    # ---- token-mixer half
    dmix = K.dgrad(dx2, att.out_proj.weight)
    _wgrad(arena, att.out_proj.weight, None, dx2, mix, wg)
    if blk.is_linear:
        dh1 = gdn_backward(att, arena, att_saved, dmix, rt, wg)
    else:
        dh1 = gated_attention_backward(att, arena, h1, att_saved, dmix, rt, wg)
    gview, gacc = _vecgrad(arena, blk.norm1.scale)
    dx, _ = K.rmsnorm_bwd(x, w1, rstd1, dh1, dres=dx2, dw_out=gview, dw_accumulate=gacc)
    _flush_wgrads(wg)
    hook = getattr(blk, "_grad_ready", None)
    if hook is not None:
        hook(blk)
    return dx


class Qwen35BlockFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, rt, keep, *params):
        B, S, d = x.shape
        x2 = x.reshape(B * S, d)
        y, saved = block_forward(blk, x2 if x2.is_contiguous() else x2.contiguous(), rt, keep)
        ctx.blk, ctx.rt, ctx.saved, ctx.shape = blk, rt, saved, (B, S, d)
        return y.view(B, S, d)

    @staticmethod
    def backward(ctx, dy):
        B, S, d = ctx.shape
        if ctx.saved is None:
            raise RuntimeError("Qwen35BlockFn: backward through a forward that ran without grad mode")
        dy2 = dy.reshape(B * S, d)
        dx = block_backward(ctx.blk, ctx.saved, dy2 if dy2.is_contiguous() else dy2.contiguous(), ctx.rt)
        ctx.saved = None
        return (dx.view(B, S, d), None, None, None) + (None,) * len(ctx.blk._param_list)


def run_block(blk, x, rt):
    if not hasattr(blk, "_param_list"):
        object.__setattr__(blk, "_param_list", list(blk.parameters()))
    L.require_gpu(x)
    if x.dtype != BF16:
        raise TypeError(f"Qwen3.5 block expects bf16 activations, got {x.dtype}")
    return Qwen35BlockFn.apply(x, blk, rt, torch.is_grad_enabled(), *blk._param_list)


class MixerFn(torch.autograd.Function):
    """A token mixer called on its own (GatedAttention / MRoPEGatedAttention / FusedGatedDeltaNet .forward): projection half +
    out_proj as one node."""

    @staticmethod
    def forward(ctx, x, att, rt, keep, *params):
        arena = arena_for_bf16(att)
        B, S, d = x.shape
        h1 = x.reshape(B * S, d)
        h1 = h1 if h1.is_contiguous() else h1.contiguous()
        if att.is_linear:
            mix, saved = gdn_forward(att, arena, h1, rt, keep)
        else:
            mix, saved = gated_attention_forward(att, arena, h1, rt)
        y = K.gemm(L.GEMM_NT, mix, att.out_proj.weight)
        ctx.att, ctx.rt, ctx.saved, ctx.shape = att, rt, (h1, mix, saved) if keep else None, (B, S, d)
        return y.view(B, S, -1)

    @staticmethod
    def backward(ctx, dy):
        att, (B, S, d) = ctx.att, ctx.shape
        if ctx.saved is None:
            raise RuntimeError("MixerFn: backward through a forward that ran without grad mode")
        arena = arena_for_bf16(att)
        h1, mix, saved = ctx.saved
        dy2 = dy.reshape(B * S, -1)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        dmix = K.dgrad(dy2, att.out_proj.weight)
        _wgrad(arena, att.out_proj.weight, None, dy2, mix)
        dh1 = gdn_backward(att, arena, saved, dmix, ctx.rt) if att.is_linear else gated_attention_backward(att, arena, h1, saved, dmix, ctx.rt)
        ctx.saved = None
        return (dh1.view(B, S, d), None, None, None) + (None,) * len(att._param_list)


def run_mixer(att, x, rt):
    if not hasattr(att, "_param_list"):
        object.__setattr__(att, "_param_list", list(att.parameters()))
    L.require_gpu(x)
    if x.dtype != BF16:
        raise TypeError(f"Qwen3.5 token mixers expect bf16 activations, got {x.dtype}")
    return MixerFn.apply(x, att, rt, torch.is_grad_enabled(), *att._param_list)


class ZCRMSNormFn(torch.autograd.Function):
    """ZeroCenteredRMSNorm on [..., width] bf16 (final_norm, or the module called on its own)."""

    @staticmethod
    def forward(ctx, x, mod, scale):
        arena_for_bf16(mod)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1])
        x2 = x2 if x2.is_contiguous() else x2.contiguous()
        w = Q.zc_weight(mod.scale)
        y, rstd = K.rmsnorm_fwd(x2, w, eps=mod.eps)
        ctx.mod, ctx.saved, ctx.shp = mod, (x2, w, rstd), shp
        return y.view(shp)

    @staticmethod
    def backward(ctx, dy):
        mod = ctx.mod
        x2, w, rstd = ctx.saved
        dy2 = dy.reshape(x2.shape)
        dy2 = dy2 if dy2.is_contiguous() else dy2.contiguous()
        gview, gacc = _vecgrad(arena_for_bf16(mod), mod.scale)
        dx, _ = K.rmsnorm_bwd(x2, w, rstd, dy2, dw_out=gview, dw_accumulate=gacc)
        return dx.view(ctx.shp), None, None


class GatedDeltaRuleFn(torch.autograd.Function):
    """gated_delta_rule(queries, keys, values, beta, alpha) with the reference's (b, h, s, d) operands
    (qwen3_next_attention.py:103-159); q/k heads already expanded to the value heads."""

    @staticmethod
    def forward(ctx, q, k, v, beta, alpha, prev_state=None):
        b, h, s, dk = q.shape
        dv = v.shape[-1]
        tm = lambda t: t.permute(0, 2, 1, 3).reshape(b * s, -1).contiguous()
        q2, k2, v2 = tm(q.to(BF16)), tm(k.to(BF16)), tm(v.to(BF16))
        be = beta.to(F32).permute(0, 2, 1).reshape(b * s, h).contiguous()
        al = alpha.to(F32).permute(0, 2, 1).reshape(b * s, h).contiguous()
        # a carried-in state (reference :103: prev_state) is copied: the kernel advances the state in place, the caller's tensor stays as it was
        state = None if prev_state is None else prev_state.detach().to(F32).contiguous().clone()
        o, ck, fin = Q.gated_delta_rule_fwd(q2, k2, v2, be, al, b, s, h, h, dk, dv, keep=True, want_state=True, state=state)
        ctx.saved = (q2, k2, v2, be, al, ck)
        ctx.meta = (b, h, s, dk, dv, q.dtype, k.dtype, v.dtype, beta.dtype, alpha.dtype, None if prev_state is None else prev_state.dtype)
        ctx.set_materialize_grads(False)  # an unused last state arrives as None, not as a zero tensor the kernel would have to read
        return o.view(b, s, h, dv).permute(0, 2, 1, 3).to(q.dtype), fin

    @staticmethod
    def backward(ctx, do, dstate):
        q2, k2, v2, be, al, ck = ctx.saved
        b, h, s, dk, dv, qd, kd, vd, bd, ad, sd = ctx.meta
        if do is None:
            do = torch.zeros((b, h, s, dv), dtype=BF16, device=q2.device)
        do2 = do.to(BF16).permute(0, 2, 1, 3).reshape(b * s, h * dv).contiguous()
        dv2 = torch.empty_like(v2)
        d_final = None if dstate is None else dstate.to(F32).contiguous()
        res = Q.gated_delta_rule_bwd(q2, k2, v2, be, al, ck, do2, dv2, b, s, h, h, dk, dv, d_final=d_final, want_d_initial=sd is not None)
        dq, dk_, dbe, dal = res[:4]
        hm = lambda t, d: t.view(b, s, h, d).permute(0, 2, 1, 3)
        hs = lambda t: t.view(b, s, h).permute(0, 2, 1)
        return hm(dq, dk).to(qd), hm(dk_, dk).to(kd), hm(dv2, dv).to(vd), hs(dbe).to(bd), hs(dal).to(ad), (res[4].to(sd) if sd is not None else None)
