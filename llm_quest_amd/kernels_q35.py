"""Tensor-level launchers for the Qwen3.5 hybrid text stack (csrc/qwen35.hip, csrc/attention_generic.hip) -- no autograd here.

Same contract as ``kernels.py``: shapes / strides / dtypes are checked on the host before a pointer reaches the GPU, outputs
are allocated with torch (device memory plumbing only), kernels are enqueued on torch's current stream.  Column blocks of a
fused projection are passed as 2-D views (unit inner stride, row pitch = the projection's width).
"""


import torch

from . import _lib as L
from .kernels import BF16, F32, NORM_PARTS, _check_attn_operand

_WS = {}


def _cols(t, name, dtype=BF16):
    if t.dim() != 2 or t.stride(1) != 1 or t.dtype != dtype:
        raise ValueError(f"{name}: expected a 2-D {dtype} tensor with unit inner stride, got {t.dtype} {tuple(t.shape)} {t.stride()}")


def _reduce_parts(part, out=None, accumulate=False):
    """out[n] (+)= sum_p part[p, n]; out fp32 / bf16 (new fp32 tensor if None)."""
    parts, n = part.shape
    if out is None:
        out = torch.empty(n, dtype=F32, device=part.device)
        accumulate = False
    L.require_gpu(part, out)
    L.call("mi355_reduce_rows_f32", parts, n, L.ptr(part), L.ptr(out), L.dt_code(out.dtype), int(accumulate))
    return out


def zc_weight(scale):
    """bf16(1 + scale) of ZeroCenteredRMSNorm."""
    L.require_gpu(scale)
    if scale.dtype != BF16 or not scale.is_contiguous():
        raise ValueError("zc_weight: scale must be contiguous bf16")
    w = torch.empty_like(scale)
    L.call("mi355_zc_weight", scale.numel(), L.ptr(scale), L.ptr(w))
    return w


def mrope_table(cos, sin, position_ids, mrope_section):
    """cos/sin fp32 [ctx, R]; position_ids int64 (3, b, s) -> per-token tables fp32 [b*s, R]."""
    L.require_gpu(cos, sin, position_ids)
    if cos.dtype != F32 or sin.dtype != F32 or not cos.is_contiguous() or not sin.is_contiguous() or cos.shape != sin.shape:
        raise ValueError("mrope_table: cos/sin must be contiguous fp32 [ctx, R]")
    if position_ids.dim() != 3 or position_ids.shape[0] != 3:
        raise ValueError(f"mrope_table: position_ids must be (3, b, s), got {tuple(position_ids.shape)}")
    ctx, R = cos.shape
    if sum(mrope_section) != R // 2:
        raise ValueError(f"mrope_table: sum(mrope_section)={sum(mrope_section)} != rotation_dim/2={R // 2}")
    pid = position_ids.to(torch.int64).contiguous()
    tokens = pid.shape[1] * pid.shape[2]
    cos_t = torch.empty((tokens, R), dtype=F32, device=cos.device)
    sin_t = torch.empty_like(cos_t)
    L.call("mi355_mrope_table", tokens, R, ctx, L.ptr(cos), L.ptr(sin), L.ptr(pid), int(mrope_section[1]), int(mrope_section[2]), L.ptr(cos_t), L.ptr(sin_t))
    return cos_t, sin_t


def rowmask(x2d, mask_u8):
    L.require_gpu(x2d, mask_u8)
    if x2d.dtype != BF16 or not x2d.is_contiguous() or mask_u8.dtype != torch.uint8 or mask_u8.numel() != x2d.shape[0]:
        raise ValueError("rowmask: x contiguous bf16 [rows, width], mask uint8 [rows]")
    y = torch.empty_like(x2d)
    L.call("mi355_rowmask", x2d.shape[0], x2d.shape[1], L.ptr(x2d), L.ptr(mask_u8.contiguous()), L.ptr(y))
    return y


def _check_heads(src, H, D, head_stride, name):
    _cols(src, name)
    if src.shape[1] < (H - 1) * head_stride + D:
        raise ValueError(f"{name}: view of width {src.shape[1]} does not cover {H} heads of {D} at stride {head_stride}")


def headnorm_rope_fwd(src, H, D, head_stride, w_eff, cos_t, sin_t, pos, eps=1e-6):
    """src: 2-D view whose column 0 is head 0 (heads every ``head_stride`` columns).  Returns (out [tokens, H*D], rstd)."""
    L.require_gpu(src, w_eff, cos_t, sin_t, pos)
    _check_heads(src, H, D, head_stride, "headnorm_rope_fwd")
    tokens = src.shape[0]
    R = 0 if cos_t is None else cos_t.shape[1]
    if w_eff.dtype != BF16 or w_eff.numel() != D or not w_eff.is_contiguous():
        raise ValueError("headnorm_rope_fwd: weight must be contiguous bf16 [D]")
    if R:
        if not (cos_t.dtype == F32 and sin_t.dtype == F32 and cos_t.is_contiguous() and sin_t.is_contiguous() and sin_t.shape == cos_t.shape):
            raise ValueError("headnorm_rope_fwd: cos/sin tables must be contiguous fp32 [rows, R]")
        if not (pos.dtype == torch.int32 and pos.numel() == tokens and pos.is_contiguous()):
            raise ValueError("headnorm_rope_fwd: pos must be contiguous int32 [tokens]")
    out = torch.empty((tokens, H * D), dtype=BF16, device=src.device)
    rstd = torch.empty((tokens, H), dtype=F32, device=src.device)
    L.call("mi355_headnorm_rope_fwd", tokens, H, D, R, L.ptr(src), src.stride(0), head_stride, L.ptr(w_eff), L.ptr(cos_t), L.ptr(sin_t), L.ptr(pos), L.ptr(out), L.ptr(rstd), eps)
    return out, rstd


def headnorm_rope_bwd(src, H, D, head_stride, w_eff, cos_t, sin_t, pos, rstd, dout, dsrc, dhead_stride):
    """Writes d(src) into the strided view ``dsrc``; returns d(w_eff) fp32 [D]."""
    L.require_gpu(src, dout, dsrc)
    _check_heads(src, H, D, head_stride, "headnorm_rope_bwd")
    _check_heads(dsrc, H, D, dhead_stride, "headnorm_rope_bwd(dsrc)")
    tokens = src.shape[0]
    R = 0 if cos_t is None else cos_t.shape[1]
    if not (dout.dtype == BF16 and dout.is_contiguous() and tuple(dout.shape) == (tokens, H * D)):
        raise ValueError("headnorm_rope_bwd: dout must be contiguous bf16 [tokens, H*D]")
    parts = max(1, min(NORM_PARTS, (tokens * H + 3) // 4))
    part = torch.empty((parts, D), dtype=F32, device=src.device)
    L.call("mi355_headnorm_rope_bwd", tokens, H, D, R, L.ptr(src), src.stride(0), head_stride, L.ptr(w_eff), L.ptr(cos_t), L.ptr(sin_t), L.ptr(pos), L.ptr(rstd),
           L.ptr(dout), L.ptr(dsrc), dsrc.stride(0), dhead_stride, L.ptr(part), parts)
    return _reduce_parts(part)


def sigmoid_gate_fwd(ctx, gate, H, D, gate_head_stride):
    L.require_gpu(ctx, gate)
    _check_heads(gate, H, D, gate_head_stride, "sigmoid_gate_fwd")
    if not (ctx.dtype == BF16 and ctx.is_contiguous() and ctx.shape[1] == H * D and ctx.shape[0] == gate.shape[0]):
        raise ValueError("sigmoid_gate_fwd: ctx must be contiguous bf16 [tokens, H*D]")
    out = torch.empty_like(ctx)
    L.call("mi355_sigmoid_gate_fwd", ctx.shape[0], H, D, L.ptr(ctx), L.ptr(gate), gate.stride(0), gate_head_stride, L.ptr(out))
    return out


def sigmoid_gate_bwd(ctx, gate, H, D, gate_head_stride, dout, dgate, dgate_head_stride):
    L.require_gpu(ctx, gate, dout, dgate)
    _check_heads(gate, H, D, gate_head_stride, "sigmoid_gate_bwd")
    _check_heads(dgate, H, D, dgate_head_stride, "sigmoid_gate_bwd(dgate)")
    if not (dout.dtype == BF16 and dout.is_contiguous() and dout.shape == ctx.shape):
        raise ValueError("sigmoid_gate_bwd: dout must be contiguous bf16 like ctx")
    dctx = torch.empty_like(ctx)
    L.call("mi355_sigmoid_gate_bwd", ctx.shape[0], H, D, L.ptr(ctx), L.ptr(gate), gate.stride(0), gate_head_stride, L.ptr(dout), L.ptr(dctx), L.ptr(dgate), dgate.stride(0), dgate_head_stride)
    return dctx


def attn_generic_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=None, scale=None):
    L.require_gpu(q, k, v, key_mask)
    _check_attn_operand(q, "q", B * S, Hq * D)
    _check_attn_operand(k, "k", B * S, Hkv * D)
    _check_attn_operand(v, "v", B * S, Hkv * D)
    if key_mask is not None and not (key_mask.dtype == torch.uint8 and key_mask.is_contiguous() and tuple(key_mask.shape) == (B, S)):
        raise ValueError("attention: key_mask must be contiguous uint8 [B,S]")
    o = torch.empty((B * S, Hq * D), dtype=BF16, device=q.device)
    lse = torch.empty((B, Hq, S), dtype=F32, device=q.device)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_generic_fwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0), L.ptr(lse), L.ptr(key_mask), scale)
    return o, lse


def attn_generic_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=None, scale=None):
    L.require_gpu(q, k, v, o, do, lse, dq, dk, dv)
    for t, n, w in ((q, "q", Hq), (k, "k", Hkv), (v, "v", Hkv), (o, "o", Hq), (do, "do", Hq), (dq, "dq", Hq), (dk, "dk", Hkv), (dv, "dv", Hkv)):
        _check_attn_operand(t, n, B * S, w * D)
    if not (lse.dtype == F32 and lse.is_contiguous() and tuple(lse.shape) == (B, Hq, S)):
        raise ValueError("attention: lse must be contiguous fp32 [B,Hq,S]")
    delta = torch.empty_like(lse)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_generic_bwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
           L.ptr(do), do.stride(0), L.ptr(lse), L.ptr(delta), L.ptr(dq), dq.stride(0), L.ptr(dk), dk.stride(0), L.ptr(dv), dv.stride(0), L.ptr(key_mask), scale)


def gdn_gates_fwd(b_lin, a_lin, log_A, dt_bias):
    L.require_gpu(b_lin, a_lin, log_A, dt_bias)
    _cols(b_lin, "gdn_gates_fwd(b_lin)")
    _cols(a_lin, "gdn_gates_fwd(a_lin)")
    tokens, Hv = b_lin.shape
    if a_lin.shape != b_lin.shape or a_lin.stride(0) != b_lin.stride(0) or log_A.dtype != F32 or dt_bias.dtype != BF16 or log_A.numel() != Hv or dt_bias.numel() != Hv:
        raise ValueError("gdn_gates_fwd: a_lin/b_lin [tokens, Hv] views of one projection, log_A fp32 [Hv], dt_bias bf16 [Hv]")
    beta = torch.empty((tokens, Hv), dtype=F32, device=b_lin.device)
    alpha = torch.empty_like(beta)
    L.call("mi355_gdn_gates_fwd", tokens, Hv, L.ptr(b_lin), L.ptr(a_lin), b_lin.stride(0), L.ptr(log_A), L.ptr(dt_bias), L.ptr(beta), L.ptr(alpha))
    return beta, alpha


def gdn_gates_bwd(b_lin, a_lin, log_A, dt_bias, dbeta, dalpha, db_lin, da_lin):
    """Writes d(b_lin), d(a_lin) into the given column views; returns (dlog_A fp32 [Hv], ddt_bias fp32 [Hv])."""
    L.require_gpu(b_lin, a_lin, dbeta, dalpha, db_lin, da_lin)
    tokens, Hv = b_lin.shape
    _cols(db_lin, "gdn_gates_bwd(db_lin)")
    _cols(da_lin, "gdn_gates_bwd(da_lin)")
    if db_lin.stride(0) != da_lin.stride(0) or tuple(db_lin.shape) != (tokens, Hv) or tuple(da_lin.shape) != (tokens, Hv):
        raise ValueError("gdn_gates_bwd: gradient views must be [tokens, Hv] columns of one buffer")
    for t in (dbeta, dalpha):
        if t.dtype != F32 or not t.is_contiguous() or tuple(t.shape) != (tokens, Hv):
            raise ValueError("gdn_gates_bwd: dbeta/dalpha must be contiguous fp32 [tokens, Hv]")
    parts = max(1, min(256, (tokens * Hv + 255) // 256))
    part = torch.empty((parts, 2 * Hv), dtype=F32, device=b_lin.device)
    L.call("mi355_gdn_gates_bwd", tokens, Hv, L.ptr(b_lin), L.ptr(a_lin), b_lin.stride(0), L.ptr(log_A), L.ptr(dt_bias), L.ptr(dbeta), L.ptr(dalpha),
           L.ptr(db_lin), L.ptr(da_lin), db_lin.stride(0), L.ptr(part), parts)
    both = _reduce_parts(part)
    return both[:Hv], both[Hv:]


def causal_conv_silu_fwd(x, w, B, S):
    """x: [B*S, C] view (row pitch = projection width); w: conv1d.weight (C, 1, k) bf16 contiguous."""
    L.require_gpu(x, w)
    _cols(x, "causal_conv_silu_fwd")
    C = x.shape[1]
    if x.shape[0] != B * S or w.dtype != BF16 or not w.is_contiguous() or w.shape[0] != C or w.numel() != C * w.shape[-1]:
        raise ValueError("causal_conv_silu_fwd: x [B*S, C], weight contiguous bf16 (C, 1, k)")
    y = torch.empty((B * S, C), dtype=BF16, device=x.device)
    L.call("mi355_causal_conv_silu_fwd", B, S, C, w.shape[-1], L.ptr(x), x.stride(0), L.ptr(w), L.ptr(y))
    return y


def causal_conv_silu_step(x_new, conv_state, w):
    """x_new [B, C] view; conv_state bf16 [B, k, C] (updated in place); returns y [B, C]."""
    L.require_gpu(x_new, conv_state, w)
    _cols(x_new, "causal_conv_silu_step")
    B, C = x_new.shape
    if not (conv_state.dtype == BF16 and conv_state.is_contiguous() and tuple(conv_state.shape) == (B, w.shape[-1], C)):
        raise ValueError("causal_conv_silu_step: conv_state must be contiguous bf16 [B, k, C]")
    y = torch.empty((B, C), dtype=BF16, device=x_new.device)
    L.call("mi355_causal_conv_silu_step", B, C, w.shape[-1], L.ptr(x_new), x_new.stride(0), L.ptr(conv_state), L.ptr(w), L.ptr(y))
    return y


CONV_TOKEN_CHUNK = 32


def causal_conv_silu_bwd(x, w, dy, dx, B, S):
    """Writes d(x) into the column view ``dx``; returns d(weight) fp32 flat [C*k]."""
    L.require_gpu(x, w, dy, dx)
    _cols(x, "causal_conv_silu_bwd")
    _cols(dx, "causal_conv_silu_bwd(dx)")
    C, ks = x.shape[1], w.shape[-1]
    if not (dy.dtype == BF16 and dy.is_contiguous() and tuple(dy.shape) == (B * S, C)) or tuple(dx.shape) != (B * S, C):
        raise ValueError("causal_conv_silu_bwd: dy contiguous bf16 [B*S, C], dx a [B*S, C] view")
    ws = torch.empty((B * S, C), dtype=BF16, device=x.device)
    chunks = B * ((S + CONV_TOKEN_CHUNK - 1) // CONV_TOKEN_CHUNK)
    part = torch.empty((chunks, C * ks), dtype=F32, device=x.device)
    L.call("mi355_causal_conv_silu_bwd", B, S, C, ks, L.ptr(x), x.stride(0), L.ptr(w), L.ptr(dy), L.ptr(ws), L.ptr(dx), dx.stride(0), L.ptr(part), CONV_TOKEN_CHUNK)
    return _reduce_parts(part)


def l2norm_fwd(x, H, D):
    L.require_gpu(x)
    _cols(x, "l2norm_fwd")
    if x.shape[1] != H * D:
        raise ValueError("l2norm_fwd: view must be [tokens, H*D]")
    y = torch.empty((x.shape[0], H * D), dtype=BF16, device=x.device)
    L.call("mi355_l2norm_fwd", x.shape[0], H, D, L.ptr(x), x.stride(0), L.ptr(y))
    return y


def l2norm_bwd(x, dy, dx, H, D):
    L.require_gpu(x, dy, dx)
    _cols(x, "l2norm_bwd")
    _cols(dx, "l2norm_bwd(dx)")
    if not (dy.dtype == BF16 and dy.is_contiguous() and tuple(dy.shape) == (x.shape[0], H * D)) or dx.shape != x.shape:
        raise ValueError("l2norm_bwd: dy contiguous bf16 [tokens, H*D], dx a view shaped like x")
    L.call("mi355_l2norm_bwd", x.shape[0], H, D, L.ptr(x), x.stride(0), L.ptr(dy), L.ptr(dx), dx.stride(0))


def gdr_chunk():
    return L.load().mi355_gated_delta_rule_chunk()


def _check_gdr(q, k, v, beta, alpha, B, S, Hqk, Hv, Dk, Dv):
    for t, n in ((q, "q"), (k, "k")):
        if not (t.dtype == BF16 and t.is_contiguous() and tuple(t.shape) == (B * S, Hqk * Dk)):
            raise ValueError(f"gated_delta_rule: {n} must be contiguous bf16 [B*S, Hqk*Dk]")
    _cols(v, "gated_delta_rule(v)")
    if tuple(v.shape) != (B * S, Hv * Dv):
        raise ValueError("gated_delta_rule: v must be a [B*S, Hv*Dv] view")
    for t, n in ((beta, "beta"), (alpha, "alpha")):
        if not (t.dtype == F32 and t.is_contiguous() and tuple(t.shape) == (B * S, Hv)):
            raise ValueError(f"gated_delta_rule: {n} must be contiguous fp32 [B*S, Hv]")


def gated_delta_rule_fwd(q, k, v, beta, alpha, B, S, Hqk, Hv, Dk, Dv, keep=True, want_state=False, state=None):
    """Returns (o bf16 [B*S, Hv*Dv], checkpoints or None, final_state or None).  ``state`` fp32 [B, Hv, Dv, Dk]: carried-in recurrent
    state, updated IN PLACE (and returned as final_state)."""
    L.require_gpu(q, k, v, beta, alpha)
    _check_gdr(q, k, v, beta, alpha, B, S, Hqk, Hv, Dk, Dv)
    o = torch.empty((B * S, Hv * Dv), dtype=BF16, device=q.device)
    ck = None
    if keep:
        ch = gdr_chunk()
        ck = torch.empty((B, Hv, (S + ch - 1) // ch, Dv, Dk), dtype=F32, device=q.device)
    if state is not None:
        L.require_gpu(state)
        if not (state.dtype == F32 and state.is_contiguous() and tuple(state.shape) == (B, Hv, Dv, Dk)):
            raise ValueError("gated_delta_rule_fwd: state must be contiguous fp32 [B, Hv, Dv, Dk]")
        fin = state
    else:
        fin = torch.empty((B, Hv, Dv, Dk), dtype=F32, device=q.device) if want_state else None
    L.call("mi355_gated_delta_rule_fwd", B, S, Hqk, Hv, Dk, Dv, L.ptr(q), L.ptr(k), L.ptr(v), v.stride(0), L.ptr(beta), L.ptr(alpha), L.ptr(o), L.ptr(ck), L.ptr(state), L.ptr(fin))
    return o, ck, fin


def gated_delta_rule_bwd(q, k, v, beta, alpha, ck, do, dv, B, S, Hqk, Hv, Dk, Dv, d_final=None, want_d_initial=False):
    """dv: destination view [B*S, Hv*Dv].  Returns (dq, dk bf16 [B*S, Hqk*Dk], dbeta, dalpha fp32 [B*S, Hv]) -- and, with ``want_d_initial``, the fp32
    [B, Hv, Dv, Dk] gradient of the carried-in state as a fifth element.  ``d_final``: the gradient arriving at the forward's final state (or None)."""
    L.require_gpu(q, k, v, beta, alpha, ck, do, dv, d_final)
    if d_final is not None and not (d_final.dtype == F32 and d_final.is_contiguous() and tuple(d_final.shape) == (B, Hv, Dv, Dk)):
        raise ValueError("gated_delta_rule_bwd: d_final must be contiguous fp32 [B, Hv, Dv, Dk]")
    _check_gdr(q, k, v, beta, alpha, B, S, Hqk, Hv, Dk, Dv)
    _cols(dv, "gated_delta_rule_bwd(dv)")
    if not (do.dtype == BF16 and do.is_contiguous() and tuple(do.shape) == (B * S, Hv * Dv)) or tuple(dv.shape) != (B * S, Hv * Dv):
        raise ValueError("gated_delta_rule_bwd: do contiguous bf16 [B*S, Hv*Dv], dv a view of that shape")
    need = L.load().mi355_gated_delta_rule_bwd_workspace_bytes(B, S, Hv, Dk, Dv)
    key = str(q.device)
    ws = _WS.get(key)
    if ws is None or ws.numel() * 4 < need:
        _WS.pop(key, None)
        ws = torch.empty((need + 3) // 4, dtype=F32, device=q.device)
        _WS[key] = ws
    dq = torch.empty_like(q)
    dk = torch.empty_like(k)
    dbeta = torch.empty_like(beta)
    dalpha = torch.empty_like(alpha)
    d_init = torch.empty((B, Hv, Dv, Dk), dtype=F32, device=q.device) if want_d_initial else None
    L.call("mi355_gated_delta_rule_bwd", B, S, Hqk, Hv, Dk, Dv, L.ptr(q), L.ptr(k), L.ptr(v), v.stride(0), L.ptr(beta), L.ptr(alpha), L.ptr(ck), L.ptr(do),
           L.ptr(dq), L.ptr(dk), L.ptr(dv), dv.stride(0), L.ptr(dbeta), L.ptr(dalpha), L.ptr(ws), ws.numel() * 4, L.ptr(d_final), L.ptr(d_init))
    return (dq, dk, dbeta, dalpha, d_init) if want_d_initial else (dq, dk, dbeta, dalpha)


def gated_rmsnorm_fwd(o, w_f32, gate, H, D, eps=1e-6):
    L.require_gpu(o, w_f32, gate)
    _cols(gate, "gated_rmsnorm_fwd(gate)")
    if not (o.dtype == BF16 and o.is_contiguous() and o.shape[1] == H * D) or tuple(gate.shape) != tuple(o.shape) or w_f32.dtype != F32 or w_f32.numel() != D:
        raise ValueError("gated_rmsnorm_fwd: o contiguous bf16 [tokens, H*D], gate a view of that shape, weight fp32 [D]")
    out = torch.empty_like(o)
    rstd = torch.empty((o.shape[0], H), dtype=F32, device=o.device)
    L.call("mi355_gated_rmsnorm_fwd", o.shape[0], H, D, L.ptr(o), L.ptr(w_f32), L.ptr(gate), gate.stride(0), L.ptr(out), L.ptr(rstd), eps)
    return out, rstd


def gated_rmsnorm_bwd(o, w_f32, gate, rstd, dout, dgate, H, D):
    """Writes d(gate) into the column view ``dgate``; returns (d_o bf16, dw fp32 [D])."""
    L.require_gpu(o, gate, rstd, dout, dgate)
    _cols(dgate, "gated_rmsnorm_bwd(dgate)")
    if not (dout.dtype == BF16 and dout.is_contiguous() and dout.shape == o.shape) or tuple(dgate.shape) != tuple(o.shape):
        raise ValueError("gated_rmsnorm_bwd: dout contiguous bf16 like o, dgate a view of that shape")
    d_o = torch.empty_like(o)
    parts = max(1, min(NORM_PARTS, (o.shape[0] * H + 3) // 4))
    part = torch.empty((parts, D), dtype=F32, device=o.device)
    L.call("mi355_gated_rmsnorm_bwd", o.shape[0], H, D, L.ptr(o), L.ptr(w_f32), L.ptr(gate), gate.stride(0), L.ptr(rstd), L.ptr(dout), L.ptr(d_o), L.ptr(dgate), dgate.stride(0), L.ptr(part), parts)
    return d_o, _reduce_parts(part)
