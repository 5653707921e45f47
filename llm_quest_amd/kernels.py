"""Tensor-level launchers over the C ABI (no autograd here).

Every launcher checks on the host that shapes / strides / dtypes are what the kernel and its grid assume before any
pointer reaches the GPU, allocates outputs with torch (device memory plumbing only) and enqueues the HIP kernel on
torch's current stream.
"""

import os

import torch

from . import _lib as L

BF16, F32 = torch.bfloat16, torch.float32
NORM_PARTS = int(os.environ.get("MI355_NORM_PARTS", "512"))  # max blocks (= partial rows) of the norm backward kernels
QK_PARTS = int(os.environ.get("MI355_QK_PARTS", "768"))  # ... of the QK-norm + RoPE backward (a wave walks one token at a time: more blocks = more rows in flight)


def _rowmajor(t, name):
    if t.dim() != 2 or t.stride(1) != 1:
        raise ValueError(f"{name}: expected a 2-D tensor with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")


import os

_TILE_OVERRIDE = int(os.environ.get("MI355_GEMM_TILE", "0"))  # profiling knob: force a GEMM tile configuration
_TILE_BY_FORM = {f: int(os.environ.get("MI355_GEMM_TILE_" + n, "0")) for f, n in ((L.GEMM_NT, "NT"), (L.GEMM_NN, "NN"), (L.GEMM_TN, "TN"))}
_TILE_SWIGLU_FWD = int(os.environ.get("MI355_GEMM_TILE_SWIGLU_FWD", "0"))  # profiling knobs: tile of the two fused SwiGLU GEMMs
_TILE_SWIGLU_BWD = int(os.environ.get("MI355_GEMM_TILE_SWIGLU_BWD", "0"))
_TILE_NT_PLAIN = int(os.environ.get("MI355_GEMM_TILE_NT_PLAIN", "0"))  # profiling knob: tile of NT launches with the plain epilogue only (the fused forms keep theirs)
DGRAD_NT = os.environ.get("MI355_DGRAD_NT", "1") != "0"  # 0: dgrad GEMMs in the NN form on the weight as stored (A/B measurements)
DGRAD_NT_MIN_ROWS = 4096  # below this the transpose pass is not paid back
_WS = {}
WS_BYTES = 512 << 20  # split-K scratch per device (fp32 slabs of the largest weight-gradient GEMM)


def _workspace(device):
    """Split-K scratch of the (device, stream) the launch goes to: two streams issuing split-K GEMMs never share slabs."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None:
        ws = torch.empty(WS_BYTES // 4, dtype=F32, device=device)
        _WS[key] = ws
    return ws


_PERSIST_USER = os.environ.get("MI355_GEMM_PERSIST_MIN_TILES")  # the user's own threshold, if any


def persistent_gemm(enabled):
    """The library takes the persistent NT kernel (one workgroup per CU for a launch's whole length, csrc/gemm.hip) by itself from 512 tiles upward.  That kernel
    assumes its 256 workgroups all START together: a CU that a collective's channel occupies takes no GEMM workgroup (each needs the CU's whole register file and
    LDS), and the workgroups that find no CU only start when the first ones have walked their whole share -- the launch then lasts up to twice as long, where the
    per-tile kernel merely loses the occupied CUs' share.  ``ddp.GradSync`` therefore switches the automatic choice off from the first collective of a backward until
    the streams have joined (``persistent_gemm(False)`` ... ``persistent_gemm(True)``); an explicit ``tile=7`` still selects it."""
    if not enabled:
        os.environ["MI355_GEMM_PERSIST_MIN_TILES"] = "1000000000"
    elif _PERSIST_USER is None:
        os.environ.pop("MI355_GEMM_PERSIST_MIN_TILES", None)
    else:
        os.environ["MI355_GEMM_PERSIST_MIN_TILES"] = _PERSIST_USER


PERSIST_MIN_TILES = 512  # csrc/gemm.hip: persist_min_tiles()


class _GemmWindow:
    """A bounded stretch of the compute stream in which NT launches stay off the persistent kernel (``ddp.GradSync`` opens one behind every bucket group it hands to the
    communication stream): ``left`` launches to go, then ``on_close`` -- the compute stream waits for the group's collectives -- and the library's own choice is back.
    ``stream``: the compute stream the window belongs to (captured when it opens); launches on other streams (a vision tower running ahead, an evaluation stream) neither
    count against it nor close it."""

    def __init__(self):
        self.left, self.on_close, self.stream, self.closed_early = 0, None, None, False
        self.stats = {"persistent_eligible": 0, "inside_window": 0, "windows": 0}


_WINDOW = _GemmWindow()


def open_gemm_window(launches, on_close=None, stream=None):
    """The next ``launches`` persistent-sized NT GEMM launches of ``stream`` (default: the current one) run on the per-tile kernel (a collective's channels hold CUs
    meanwhile; the persistent kernel's 256 workgroups must all start together, see ``persistent_gemm``); in front of the launch after them ``on_close()`` runs and the
    persistent kernel is allowed again.  A window opened while one is open extends it; both hand-offs run at its end."""
    w = _WINDOW
    if w.left == 0 and w.on_close is None:
        persistent_gemm(False)
        w.stats["windows"] += 1
        w.stream = stream
    w.left = max(w.left, int(launches))
    if on_close is not None:
        prev = w.on_close
        w.on_close = on_close if prev is None else (lambda: (prev(), on_close()))
    elif w.on_close is None:
        w.on_close = lambda: None


def close_gemm_window():
    """End the window now (``GradSync.finish_step``; also the first launch behind a window's last one)."""
    w = _WINDOW
    if w.left == 0 and w.on_close is None:
        return
    w.closed_early = w.left > 0  # ended by finish_step before its launches ran out (the hand-off callbacks may ask)
    cb, w.left, w.on_close, w.stream = w.on_close, 0, None, None
    if cb is not None:
        cb()
    persistent_gemm(True)


def _persist_min_tiles():
    return int(_PERSIST_USER) if _PERSIST_USER else PERSIST_MIN_TILES


def _nt_tick(M, N, eligible=True, device=None):
    """Called in front of every NT launch.  Counts only launches the library would put on the persistent kernel (csrc/gemm.hip ``persist_ok``: bf16 output, no bias,
    plain / SwiGLU epilogue, no tile hint -- the caller says so through ``eligible`` -- and at least the threshold's tiles), and only those of the window's own stream."""
    if not eligible or ((M + 255) // 256) * ((N + 255) // 256) < _persist_min_tiles():
        return
    w = _WINDOW
    if w.stream is not None and device is not None and torch.cuda.current_stream(device) != w.stream:
        return
    if w.left > 0:
        w.left -= 1
        w.stats["inside_window"] += 1
        return
    if w.on_close is not None:
        close_gemm_window()
    w.stats["persistent_eligible"] += 1


def gemm(form, a, b, out=None, out_dtype=BF16, bias=None, residual=None, gelu=False, allow_split_k=True, tile=0, split3_out=False):
    """C = epi(op(A) op(B) + bias) + residual.  form NT: A[M,K] B[N,K]; NN: A[M,K] B[K,N]; TN: A[K,M] B[K,N].
    ``split3_out`` (NT, no residual): the fp32 result leaves the epilogue as bf16 [M, 3N] = [hi | lo | hi] -- the activation operand of the next fp32-grade
    product (``split3``) without the fp32 round trip through memory."""
    if split3_out:
        L.require_gpu(a, b, bias)
        if form != L.GEMM_NT or residual is not None or out is not None or a.dtype != BF16 or b.dtype != BF16 or a.shape[1] != b.shape[1] or b.shape[0] % 8:
            raise ValueError("gemm(split3_out): NT form on bf16 operands, N a multiple of 8, no residual, no preallocated output")
        M, N, Kd = a.shape[0], b.shape[0], a.shape[1]
        _rowmajor(a, "A")
        _rowmajor(b, "B")
        out3 = torch.empty((M, 3 * N), dtype=BF16, device=a.device)
        L.call("mi355_gemm_bf16", form, M, N, Kd, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out3), out3.stride(0), L.DT_SPLIT3, L.ptr(bias), None, 0,
               L.EPI_GELU if gelu else L.EPI_NONE, None, 0, tile)
        return out3
    L.require_gpu(a, b, out, bias, residual)
    _rowmajor(a, "A")
    _rowmajor(b, "B")
    if a.dtype != BF16 or b.dtype != BF16:
        raise TypeError("gemm operands must be bf16")
    if form == L.GEMM_NT:
        M, K = a.shape
        N, K2 = b.shape
    elif form == L.GEMM_NN:
        M, K = a.shape
        K2, N = b.shape
    else:
        K, M = a.shape
        K2, N = b.shape
    if K != K2:
        raise ValueError(f"gemm: inner dimensions differ ({K} vs {K2})")
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=a.device)
    _rowmajor(out, "C")
    if tuple(out.shape) != (M, N):
        raise ValueError(f"gemm: output shape {tuple(out.shape)} != {(M, N)}")
    if bias is not None and (bias.dtype != F32 or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("gemm: bias must be contiguous fp32 [N]")
    ldr = 0
    if residual is not None:
        _rowmajor(residual, "residual")
        if residual.dtype != out.dtype or tuple(residual.shape) != (M, N):
            raise ValueError("gemm: residual must match the output's shape and dtype")
        ldr = residual.stride(0)
    tile_eff = tile or (_TILE_NT_PLAIN if (form == L.GEMM_NT and not gelu and M >= 4096) else 0) or _TILE_BY_FORM[form] or _TILE_OVERRIDE
    if form == L.GEMM_NT:
        _nt_tick(M, N, eligible=out.dtype == BF16 and bias is None and not gelu and tile_eff == 0 and K % 64 == 0 and K >= 128, device=a.device)
    L.call(
        "mi355_gemm_bf16", form, M, N, K, L.ptr(a), a.stride(0), L.ptr(b), b.stride(0), L.ptr(out), out.stride(0),
        L.dt_code(out.dtype), L.ptr(bias), L.ptr(residual), ldr, L.EPI_GELU if gelu else L.EPI_NONE,
        L.ptr(_workspace(a.device)) if allow_split_k else None, WS_BYTES if allow_split_k else 0,
        tile_eff,
    )
    return out


def gemm_grouped(form, problems, tile=0):
    """One launch for several independent GEMMs of one form: problems = [(a, b, out, residual_or_None), ...] (<= 8, same
    output dtype, outputs preallocated).  Used for the weight gradients of a transformer block."""
    import ctypes as _c

    if not problems:
        return
    if len(problems) > 8:
        for i in range(0, len(problems), 8):
            gemm_grouped(form, problems[i : i + 8], tile)
        return
    table = (L.GemmProblem * len(problems))()
    odt = problems[0][2].dtype
    for q, (a, b, out, residual) in zip(table, problems):
        L.require_gpu(a, b, out, residual)
        _rowmajor(a, "A")
        _rowmajor(b, "B")
        _rowmajor(out, "C")
        if a.dtype != BF16 or b.dtype != BF16:
            raise TypeError("gemm operands must be bf16")
        if form == L.GEMM_NT:
            (M, Kd), (N, K2) = a.shape, b.shape
        elif form == L.GEMM_NN:
            (M, Kd), (K2, N) = a.shape, b.shape
        else:
            (Kd, M), (K2, N) = a.shape, b.shape
        if Kd != K2 or tuple(out.shape) != (M, N) or out.dtype != odt:
            raise ValueError(f"gemm_grouped: inconsistent problem {tuple(a.shape)} x {tuple(b.shape)} -> {tuple(out.shape)} {out.dtype}")
        q.M, q.N, q.K = M, N, Kd
        q.A, q.lda, q.B, q.ldb, q.C, q.ldc = a.data_ptr(), a.stride(0), b.data_ptr(), b.stride(0), out.data_ptr(), out.stride(0)
        q.residual, q.ldr = None, 0
        if residual is not None:
            _rowmajor(residual, "residual")
            if residual.dtype != odt or tuple(residual.shape) != (M, N):
                raise ValueError("gemm_grouped: residual must match the output's shape and dtype")
            q.residual, q.ldr = residual.data_ptr(), residual.stride(0)
    L.call("mi355_gemm_bf16_grouped", form, len(problems), _c.cast(table, _c.c_void_p), L.dt_code(odt), tile or _TILE_BY_FORM[form] or (_TILE_OVERRIDE if _TILE_OVERRIDE in (1, 3, 4, 5) else 0))


# Measured (ViT-B/16 training step, B = 256, same box): fused 65.5 ms, separate kernels 64.1 ms -- erf / tanh in the epilogue sit on the tile's critical
# path with one workgroup per CU, the separate kernels run HBM-bound over the whole chip.  So the separate form is the default; 1 selects the fused one.
_FUSE_GELU = os.environ.get("MI355_FUSE_GELU", "0") != "0"


def gemm_gelu_dual(x, w, bias=None, tanh=False, tile=0):
    """(y1 = x @ w^T + bias, gelu(y1)) in ONE launch (replaces gemm + gelu_fwd in a training FFN; y1 is kept for the backward)."""
    if not _FUSE_GELU:
        y1 = gemm(L.GEMM_NT, x, w, bias=bias, tile=tile)
        return y1, gelu_fwd(y1, tanh=tanh)
    L.require_gpu(x, w, bias)
    _rowmajor(x, "X")
    _rowmajor(w, "W")
    M, Kd = x.shape
    N = w.shape[0]
    if x.dtype != BF16 or w.dtype != BF16 or w.shape[1] != Kd or N % 8:
        raise ValueError("gemm_gelu_dual: bf16 X [M, K], W [N, K], N % 8 == 0")
    if bias is not None and (bias.dtype != F32 or bias.numel() != N or not bias.is_contiguous()):
        raise ValueError("gemm_gelu_dual: bias must be contiguous fp32 [N]")
    y1 = torch.empty((M, N), dtype=BF16, device=x.device)
    act = torch.empty_like(y1)
    L.call("mi355_gemm_bf16", L.GEMM_NT, M, N, Kd, L.ptr(x), x.stride(0), L.ptr(w), w.stride(0), L.ptr(y1), y1.stride(0), L.DT_BF16, L.ptr(bias),
           L.ptr(act), act.stride(0), L.EPI_GELU_DUAL_TANH if tanh else L.EPI_GELU_DUAL_ERF, None, 0, tile or _TILE_BY_FORM[L.GEMM_NT] or _TILE_OVERRIDE)
    return y1, act


def gemm_dgrad_gelu_bwd(dy, w, y1, tanh=False, tile=0):
    """d(y1) = (dy @ w) * gelu'(y1) in ONE launch (replaces the dgrad gemm + gelu_bwd); w = the following Linear's weight [N_out, F]."""
    if not _FUSE_GELU:
        return gelu_bwd(y1, dgrad(dy, w) if not tile else gemm(L.GEMM_NN, dy, w, tile=tile), tanh=tanh)  # (dgrad: the NT form on W^T above 4 096 rows, same bits)
    L.require_gpu(dy, w, y1)
    _rowmajor(dy, "dY")
    _rowmajor(w, "W")
    M, Kd = dy.shape
    F = w.shape[1]
    if dy.dtype != BF16 or w.dtype != BF16 or y1.dtype != BF16 or w.shape[0] != Kd or not y1.is_contiguous() or tuple(y1.shape) != (M, F) or F % 8:
        raise ValueError("gemm_dgrad_gelu_bwd: dY [M, N_out], W [N_out, F], pre-activation contiguous [M, F], all bf16")
    out = torch.empty_like(y1)
    L.call("mi355_gemm_bf16", L.GEMM_NN, M, F, Kd, L.ptr(dy), dy.stride(0), L.ptr(w), w.stride(0), L.ptr(out), out.stride(0), L.DT_BF16, None,
           L.ptr(y1), y1.stride(0), L.EPI_GELU_BWD_TANH if tanh else L.EPI_GELU_BWD_ERF, None, 0, tile or _TILE_BY_FORM[L.GEMM_NN] or _TILE_OVERRIDE)
    return out


def gemm_gateup_swiglu(x, w_fused, tile=0):
    """(gu [M, 2F], a [M, F]) of a SwiGLU FFN in ONE launch: gu = x @ [lin1 | lin_gate]^T as usual, a = u * silu(g) from the epilogue
    (replaces gemm + swiglu_fwd)."""
    L.require_gpu(x, w_fused)
    _rowmajor(x, "X")
    _rowmajor(w_fused, "W")
    M, Kd = x.shape
    N = w_fused.shape[0]
    if x.dtype != BF16 or w_fused.dtype != BF16 or w_fused.shape[1] != Kd or N % 64:
        raise ValueError("gemm_gateup_swiglu: bf16 X [M, K], fused weight [2F, K] with F % 32 == 0")
    gu = torch.empty((M, N), dtype=BF16, device=x.device)
    a = torch.empty((M, N // 2), dtype=BF16, device=x.device)
    _nt_tick(M, N, eligible=not (tile or _TILE_SWIGLU_FWD or _TILE_BY_FORM[L.GEMM_NT] or _TILE_OVERRIDE) and Kd % 64 == 0 and Kd >= 128, device=x.device)
    L.call("mi355_gemm_bf16", L.GEMM_NT, M, N, Kd, L.ptr(x), x.stride(0), L.ptr(w_fused), w_fused.stride(0), L.ptr(gu), gu.stride(0), L.DT_BF16, None,
           L.ptr(a), a.stride(0), L.EPI_SWIGLU_FWD, None, 0, tile or _TILE_SWIGLU_FWD or _TILE_BY_FORM[L.GEMM_NT] or _TILE_OVERRIDE)
    return gu, a


def gemm_dgrad_swiglu_bwd(dy, w, gu, tile=0):
    """d(gate-up) [M, 2F] of a SwiGLU FFN in ONE launch: d(act) = dy @ w (NN, w = lin2.weight [N_out, F]) stays in the accumulators and the
    activation's backward runs in the epilogue against ``gu`` = the forward's [lin1 | lin_gate] output (replaces gemm + swiglu_bwd)."""
    L.require_gpu(dy, w, gu)
    _rowmajor(dy, "dY")
    _rowmajor(w, "W")
    M, Kd = dy.shape
    F = w.shape[1]
    if dy.dtype != BF16 or w.dtype != BF16 or gu.dtype != BF16 or w.shape[0] != Kd or not gu.is_contiguous() or tuple(gu.shape) != (M, 2 * F):
        raise ValueError("gemm_dgrad_swiglu_bwd: dY [M, N_out], W [N_out, F], gate-up output contiguous [M, 2F], all bf16")
    out = torch.empty_like(gu)
    if DGRAD_NT and M >= DGRAD_NT_MIN_ROWS:
        wt = transpose(w)  # [F, N_out]: the same product in the K-contiguous form
        _nt_tick(M, F, eligible=not (tile or _TILE_SWIGLU_BWD or _TILE_BY_FORM[L.GEMM_NT] or _TILE_OVERRIDE) and Kd % 64 == 0 and Kd >= 128, device=dy.device)
        L.call("mi355_gemm_bf16", L.GEMM_NT, M, F, Kd, L.ptr(dy), dy.stride(0), L.ptr(wt), wt.stride(0), L.ptr(out), out.stride(0), L.DT_BF16, None,
               L.ptr(gu), gu.stride(0), L.EPI_SWIGLU_BWD, None, 0, tile or _TILE_SWIGLU_BWD or _TILE_BY_FORM[L.GEMM_NT] or _TILE_OVERRIDE)
        return out
    L.call("mi355_gemm_bf16", L.GEMM_NN, M, F, Kd, L.ptr(dy), dy.stride(0), L.ptr(w), w.stride(0), L.ptr(out), out.stride(0), L.DT_BF16, None,
           L.ptr(gu), gu.stride(0), L.EPI_SWIGLU_BWD, None, 0, tile or _TILE_BY_FORM[L.GEMM_NN] or _TILE_OVERRIDE)
    return out


def colsum(x, out=None, accumulate=False):
    """out[n] (+)= sum_m x[m, n]; x bf16 or fp32 (row-strided views allowed), out fp32."""
    L.require_gpu(x, out)
    _rowmajor(x, "X")
    if out is None:
        out = torch.empty(x.shape[1], dtype=F32, device=x.device)
    if out.dtype != F32 or out.numel() != x.shape[1] or not out.is_contiguous():
        raise ValueError("colsum: out must be contiguous fp32 [N]")
    L.call("mi355_colsum", x.shape[0], x.shape[1], L.ptr(x), L.dt_code(x.dtype), x.stride(0), L.ptr(out), int(accumulate))
    return out


def rmsnorm_fwd(x2d, w, eps=1e-6, want_rstd=True):
    L.require_gpu(x2d, w)
    if x2d.dtype != BF16 or w.dtype != BF16 or not x2d.is_contiguous() or w.numel() != x2d.shape[1]:
        raise ValueError("rmsnorm_fwd: x must be contiguous bf16 [rows,width], w bf16 [width]")
    y = torch.empty_like(x2d)
    rstd = torch.empty(x2d.shape[0], dtype=F32, device=x2d.device) if want_rstd else None
    L.call("mi355_rmsnorm_fwd", x2d.shape[0], x2d.shape[1], L.ptr(x2d), L.ptr(w), L.ptr(y), L.ptr(rstd), eps)
    return y, rstd


def rmsnorm_bwd(x2d, w, rstd, dy, dres=None, dw_out=None, dw_accumulate=False):
    """Returns (dx [+ dres], dw).  dw goes to ``dw_out`` (bf16/fp32, optionally accumulated) or a new fp32 tensor."""
    L.require_gpu(x2d, w, rstd, dy, dres)
    rows, width = x2d.shape
    if not (dy.is_contiguous() and dy.shape == x2d.shape and dy.dtype == BF16):
        raise ValueError("rmsnorm_bwd: dy must be contiguous bf16 like x")
    if dres is not None and not (dres.is_contiguous() and dres.shape == x2d.shape and dres.dtype == BF16):
        raise ValueError("rmsnorm_bwd: dres must be contiguous bf16 like x")
    dx = torch.empty_like(x2d)
    parts = min(NORM_PARTS, (rows + 3) // 4)
    part = torch.empty((parts, width), dtype=F32, device=x2d.device)
    L.call("mi355_rmsnorm_bwd", rows, width, L.ptr(x2d), L.ptr(w), L.ptr(rstd), L.ptr(dy), L.ptr(dres), L.ptr(dx), L.ptr(part), parts)
    dw = torch.empty(width, dtype=F32, device=x2d.device) if dw_out is None else dw_out
    L.call("mi355_reduce_rows_f32", parts, width, L.ptr(part), L.ptr(dw), L.dt_code(dw.dtype), int(dw_accumulate and dw_out is not None))
    return dx, dw


def qknorm_rope_fwd(qkv, qw, kw, cos, sin, pos, Hq, Hkv, D, eps=1e-6):
    """qw = kw = None: RoPE only (no per-head normalisation); rstd is then None."""
    L.require_gpu(qkv, qw, kw, cos, sin, pos)
    tokens = qkv.shape[0]
    if not (qkv.is_contiguous() and qkv.dtype == BF16 and qkv.shape[1] == (Hq + 2 * Hkv) * D):
        raise ValueError("qknorm_rope_fwd: qkv must be contiguous bf16 [tokens,(Hq+2Hkv)*D]")
    if not (cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape[1] == D):
        raise ValueError("qknorm_rope_fwd: cos/sin must be contiguous fp32 [ctx,D]")
    if not (pos.dtype == torch.int32 and pos.numel() == tokens and pos.is_contiguous()):
        raise ValueError("qknorm_rope_fwd: pos must be contiguous int32 [tokens]")
    q = torch.empty((tokens, Hq * D), dtype=BF16, device=qkv.device)
    k = torch.empty((tokens, Hkv * D), dtype=BF16, device=qkv.device)
    rstd = torch.empty((tokens, Hq + Hkv), dtype=F32, device=qkv.device) if qw is not None else None
    L.call("mi355_qknorm_rope_fwd", tokens, Hq, Hkv, D, L.ptr(qkv), L.ptr(qw), L.ptr(kw), L.ptr(cos), L.ptr(sin), L.ptr(pos), L.ptr(q), L.ptr(k), L.ptr(rstd), eps)
    return q, k, rstd


def qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, dq, dk, dqkv, Hq, Hkv, D):
    """Writes d(qkv)[:, :(Hq+Hkv)*D] into dqkv (the V third is left to the caller); returns (dqw_f32, dkw_f32).
    dq=None: the key heads only (the query heads' share ran inside attn_bwd_qnorm); dqw_f32 is then zeros."""
    L.require_gpu(qkv, dk, dqkv)
    tokens = qkv.shape[0]
    for t, n in ((dq, Hq * D), (dk, Hkv * D)):
        if t is None:
            continue
        L.require_gpu(t)
        if not (t.is_contiguous() and t.dtype == BF16 and tuple(t.shape) == (tokens, n)):
            raise ValueError("qknorm_rope_bwd: dq/dk must be contiguous bf16 [tokens,H*D]")
    if not (dqkv.is_contiguous() and dqkv.shape == qkv.shape and dqkv.dtype == BF16):
        raise ValueError("qknorm_rope_bwd: dqkv must be contiguous bf16 like qkv")
    hpw = 4 if D == 128 else 8
    parts = min(QK_PARTS, (tokens * ((Hq + Hkv + hpw - 1) // hpw) + 3) // 4)
    part = torch.empty((parts, 2 * D), dtype=F32, device=qkv.device)
    L.call("mi355_qknorm_rope_bwd", tokens, Hq, Hkv, D, L.ptr(qkv), L.ptr(qw), L.ptr(kw), L.ptr(cos), L.ptr(sin), L.ptr(pos), L.ptr(rstd), L.ptr(dq), L.ptr(dk), L.ptr(dqkv), L.ptr(part), parts)
    dw = torch.empty(2 * D, dtype=F32, device=qkv.device)
    L.call("mi355_reduce_rows_f32", parts, 2 * D, L.ptr(part), L.ptr(dw), L.DT_F32, 0)
    return dw[:D], dw[D:]


def swiglu_fwd(gu, F):
    L.require_gpu(gu)
    if not (gu.is_contiguous() and gu.dtype == BF16 and gu.shape[1] == 2 * F):
        raise ValueError("swiglu_fwd: gu must be contiguous bf16 [tokens,2F]")
    a = torch.empty((gu.shape[0], F), dtype=BF16, device=gu.device)
    L.call("mi355_swiglu_fwd", gu.shape[0], F, L.ptr(gu), L.ptr(a))
    return a


def swiglu_bwd(gu, da, F):
    L.require_gpu(gu, da)
    if not (da.is_contiguous() and da.dtype == BF16 and tuple(da.shape) == (gu.shape[0], F)):
        raise ValueError("swiglu_bwd: da must be contiguous bf16 [tokens,F]")
    dgu = torch.empty_like(gu)
    L.call("mi355_swiglu_bwd", gu.shape[0], F, L.ptr(gu), L.ptr(da), L.ptr(dgu))
    return dgu


def _check_attn_operand(t, name, tokens, width):
    if t.dtype != BF16 or t.dim() != 2 or t.stride(1) != 1 or t.shape[0] != tokens or t.shape[1] != width:
        raise ValueError(f"attention: {name} must be bf16 [tokens={tokens}, {width}] with unit inner stride, got {tuple(t.shape)} {t.stride()}")


_ATTN_ABLATE = int(os.environ.get("MI355_ATTN_ABLATE", "0")) << 8  # profiling switches of attention.hip


_ATTN_DS_SPILL = os.environ.get("MI355_ATTN_DS_SPILL", "1") != "0"  # 0: the dQ pass recomputes S and dP (no scratch)
# The dS scratch grows with S^2 (2 bytes per (query, key) of the padded square per head): 1.2 GB at the headline shape, 17 GB at S = 8192, B = 8.
# Above this cap -- or a quarter of the device's free memory, or when the allocation fails -- the backward takes the recompute form (three
# products in the dQ pass, O(S) memory, bit-identical gradients): long-context training keeps working, only slower per layer.
_ATTN_DS_SPILL_MAX = int(float(os.environ.get("MI355_ATTN_DS_SPILL_MAX_MB", "4096")) * (1 << 20))
_ATTN_WS = {}  # (device, stream handle) -> uint8 buffer; at most _ATTN_WS_SLOTS entries (least recently used dropped: dead streams do not pin memory)
_ATTN_WS_SLOTS = 4
attn_bwd_form = {"spill": 0, "recompute": 0}  # how many backward calls took each form (tests, diagnostics)


def release_attention_scratch():
    """Drop every cached dS scratch buffer (they are re-created on demand)."""
    _ATTN_WS.clear()


def _attn_scratch(device, nbytes):
    """dS scratch of the attention backward, one per (device, stream): every layer of a model reuses it (a backward pass fills it and
    empties it again before the next one starts on the same stream).  Returns None when the request is over the cap or cannot be
    allocated -- the caller then runs the recompute form.  A buffer more than 4x the request is replaced by a fitting one."""
    if nbytes > _ATTN_DS_SPILL_MAX:
        return None
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    ws = _ATTN_WS.pop(key, None)
    if ws is not None and (ws.numel() < nbytes or ws.numel() > 4 * nbytes):
        ws = None
    if ws is None:
        free, _ = torch.cuda.mem_get_info(device)
        cached = torch.cuda.memory_reserved(device) - torch.cuda.memory_allocated(device)
        if nbytes > (free + cached) // 4:
            return None
        try:
            ws = torch.empty(nbytes, dtype=torch.uint8, device=device)
        except torch.OutOfMemoryError:
            return None
    _ATTN_WS[key] = ws  # re-inserted last: dict order is the LRU order
    while len(_ATTN_WS) > _ATTN_WS_SLOTS:
        _ATTN_WS.pop(next(iter(_ATTN_WS)))
    return ws


def attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=None, causal=True, scale=None):
    """q [B*S,Hq*D], k/v [B*S,Hkv*D] (row-strided views allowed).  Returns (o [B*S,Hq*D], lse fp32 [B,Hq,S])."""
    L.require_gpu(q, k, v, key_mask)
    _check_attn_operand(q, "q", B * S, Hq * D)
    _check_attn_operand(k, "k", B * S, Hkv * D)
    _check_attn_operand(v, "v", B * S, Hkv * D)
    if key_mask is not None and not (key_mask.dtype == torch.uint8 and key_mask.is_contiguous() and tuple(key_mask.shape) == (B, S)):
        raise ValueError("attention: key_mask must be contiguous uint8 [B,S]")
    o = torch.empty((B * S, Hq * D), dtype=BF16, device=q.device)
    lse = torch.empty((B, Hq, S), dtype=F32, device=q.device)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_fwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0), L.ptr(lse), L.ptr(key_mask), int(causal) | _ATTN_ABLATE, scale)
    return o, lse


def attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=None, causal=True, scale=None):
    """dq/dk/dv are caller-provided (possibly row-strided) destinations."""
    L.require_gpu(q, k, v, o, do, lse, dq, dk, dv)
    for t, n, w in ((q, "q", Hq), (k, "k", Hkv), (v, "v", Hkv), (o, "o", Hq), (do, "do", Hq), (dq, "dq", Hq), (dk, "dk", Hkv), (dv, "dv", Hkv)):
        _check_attn_operand(t, n, B * S, w * D)
    if not (lse.dtype == F32 and lse.is_contiguous() and tuple(lse.shape) == (B, Hq, S)):
        raise ValueError("attention: lse must be contiguous fp32 [B,Hq,S]")
    delta = torch.empty_like(lse)
    scale = D ** -0.5 if scale is None else scale
    ws, need = None, L.load().mi355_attn_bwd_workspace_bytes(B, S, Hq, D) if _ATTN_DS_SPILL else 0
    if need:
        ws = _attn_scratch(q.device, need)
        if ws is None:
            need = 0  # over the cap / no memory: the recompute form (mi355_attn_bwd_ws with a NULL workspace)
    attn_bwd_form["spill" if need else "recompute"] += 1
    L.call(
        "mi355_attn_bwd_ws", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
        L.ptr(do), do.stride(0), L.ptr(lse), L.ptr(delta), L.ptr(dq), dq.stride(0), L.ptr(dk), dk.stride(0), L.ptr(dv), dv.stride(0),
        L.ptr(key_mask), int(causal) | _ATTN_ABLATE, scale, L.ptr(ws), need,
    )


FUSE_QNORM_BWD = os.environ.get("MI355_FUSE_QNORM_BWD", "1") != "0"  # A/B knob: 0 = dQ matrix + the separate QK-norm / RoPE backward for every head


_ROPE_CS16 = {}  # (cos.data_ptr, sin.data_ptr, version counters) -> compact bf16 table, or None when the table's halves differ
ROPE_CS16 = os.environ.get("MI355_ROPE_CS16", "1") != "0"  # A/B knob: 0 = the dQ write-out reads the fp32 cos / sin tables


def rope_cs16(cos, sin):
    """[positions][cos[:, :D/2] | sin[:, :D/2]] in bf16 for the fused dQ write-out, or None when the fp32 tables' halves are not identical
    (checked once per table on the device; the result is cached per table storage and version)."""
    if not ROPE_CS16:
        return None
    key = (cos.data_ptr(), sin.data_ptr(), cos._version, sin._version, tuple(cos.shape))
    hit = _ROPE_CS16.get(key, False)
    if hit is not False:
        return hit
    h = cos.shape[-1] // 2
    same = bool(torch.equal(cos[..., :h], cos[..., h:]) and torch.equal(sin[..., :h], sin[..., h:]))  # one host sync per table, at first use
    tab = torch.cat((cos[..., :h], sin[..., :h]), dim=-1).to(BF16).contiguous() if same else None
    if len(_ROPE_CS16) > 16:
        _ROPE_CS16.clear()
    _ROPE_CS16[key] = tab
    return tab


FUSE_ATTN_DELTA = os.environ.get("MI355_FUSE_ATTN_DELTA", "1") != "0"  # A/B knob: 0 = plain out-projection dgrad + the delta pass inside the attention backward


def dgrad_attn_delta(dy, w, ctx, lse, B, S, Hq, D):
    """d(ctx) = dy @ w (the out-projection's dgrad, NT on the transposed weight) whose epilogue also leaves the attention backward's row constants
    (delta = rowsum(d(ctx) * ctx) per head, -delta, -lse * log2 e) in the backward's scratch: the stand-alone delta pass (1 read of ctx + 1 of
    d(ctx)) disappears.  Returns (dctx, delta) -- pass ``delta`` on to ``attn_bwd_qnorm`` -- or None when the form does not apply (head_dim != 128,
    few rows, no scratch, knob off): the caller then runs ``dgrad``."""
    M = dy.shape[0]
    if not (FUSE_ATTN_DELTA and FUSE_QNORM_BWD and _ATTN_DS_SPILL and DGRAD_NT and D == 128 and M == B * S and M >= max(256, DGRAD_NT_MIN_ROWS)):
        return None
    L.require_gpu(dy, w, ctx, lse)
    _rowmajor(dy, "dY")
    _rowmajor(w, "W")
    _rowmajor(ctx, "ctx")
    N, Kd = Hq * D, dy.shape[1]
    if not (dy.dtype == BF16 and w.dtype == BF16 and ctx.dtype == BF16 and tuple(w.shape) == (Kd, N) and tuple(ctx.shape) == (M, N) and (Kd & 7) == 0
            and lse.dtype == F32 and lse.is_contiguous() and tuple(lse.shape) == (B, Hq, S)):
        raise ValueError("dgrad_attn_delta: dY bf16 [B*S, d], W bf16 [d, Hq*D], ctx bf16 [B*S, Hq*D], lse contiguous fp32 [B, Hq, S]")
    lib = L.load()
    need = lib.mi355_attn_bwd_workspace_bytes(B, S, Hq, D)
    ws = _attn_scratch(dy.device, need) if need else None
    if ws is None:
        return None
    off0, off1 = (lib.mi355_attn_bwd_workspace_rowconst_offset(B, S, Hq, D, i) for i in (0, 1))
    wt = transpose(w)  # [Hq*D, d]
    L.require_gpu(dy, wt, ctx, lse)
    out = torch.empty((M, N), dtype=BF16, device=dy.device)
    delta = torch.empty_like(lse)
    _nt_tick(M, N, eligible=Kd % 64 == 0 and Kd >= 128, device=dy.device)
    L.call("mi355_gemm_bf16_attn_delta", M, N, Kd, L.ptr(dy), dy.stride(0), L.ptr(wt), wt.stride(0), L.ptr(out), out.stride(0), L.ptr(ctx), ctx.stride(0),
           S, Hq, D, L.ptr(lse), L.ptr(delta), ws.data_ptr() + off0, ws.data_ptr() + off1)
    return out, delta


def attn_bwd_qnorm(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dk, dv, qkv, qw, cos, sin, pos, rstd, dqkv, key_mask=None, causal=True, scale=None, delta=None):
    """attn_bwd whose dQ pass ends in the backward of the QK-norm + RoPE of the query heads: writes d(qkv)[:, :Hq*D] into ``dqkv`` (dQ never exists
    as a matrix) and returns the fp32 [D] gradient of the query norm weight -- or None when this form does not apply (head_dim != 128, no scratch
    for the one-product dQ pass, knob off): the caller then runs attn_bwd + qknorm_rope_bwd."""
    if not (FUSE_QNORM_BWD and _ATTN_DS_SPILL and D == 128) or qw is None:
        return None
    L.require_gpu(q, k, v, o, do, lse, dk, dv, qkv, qw, cos, sin, pos, rstd, dqkv)
    for t, n, w in ((q, "q", Hq), (k, "k", Hkv), (v, "v", Hkv), (o, "o", Hq), (do, "do", Hq), (dk, "dk", Hkv), (dv, "dv", Hkv)):
        _check_attn_operand(t, n, B * S, w * D)
    if not (lse.dtype == F32 and lse.is_contiguous() and tuple(lse.shape) == (B, Hq, S)):
        raise ValueError("attention: lse must be contiguous fp32 [B,Hq,S]")
    H = rstd.shape[1]
    if not (qkv.dtype == BF16 and dqkv.dtype == BF16 and qkv.shape == dqkv.shape and qkv.shape[0] == B * S and qkv.stride(1) == 1 and dqkv.stride(1) == 1
            and rstd.dtype == F32 and rstd.is_contiguous() and rstd.shape[0] == B * S and H >= Hq and qw.dtype == BF16 and qw.numel() == D
            and cos.dtype == F32 and sin.dtype == F32 and cos.is_contiguous() and sin.is_contiguous() and cos.shape[-1] == D and pos.dtype == torch.int32 and pos.numel() == B * S):
        raise ValueError("attn_bwd_qnorm: qkv / dqkv bf16 [B*S, >=Hq*D], rstd fp32 [B*S, heads], q weight bf16 [D], cos / sin fp32 [positions, D], pos int32 [B*S]")
    need = L.load().mi355_attn_bwd_workspace_bytes(B, S, Hq, D)
    ws = _attn_scratch(q.device, need) if need else None
    if ws is None:
        return None
    attn_bwd_form["spill"] += 1
    ready = 0
    if delta is not None:  # written, with the scratch's two row-constant arrays, by dgrad_attn_delta: the delta pass is skipped
        if not (delta.dtype == F32 and delta.is_contiguous() and delta.shape == lse.shape):
            raise ValueError("attn_bwd_qnorm: delta must be contiguous fp32 [B,Hq,S]")
        L.require_gpu(q, delta)
        ready = L.ATTN_DELTA_READY
    else:
        delta = torch.empty_like(lse)
    scale = D ** -0.5 if scale is None else scale
    cs16 = rope_cs16(cos, sin)  # plain RoPE tables: compact bf16 coefficients for the write-out (same bits, a quarter of the loads)
    L.require_gpu(q, cs16)
    parts = L.load().mi355_attn_bwd_qnorm_partials(B, S, Hq)
    part = torch.empty((parts, D), dtype=F32, device=q.device)
    L.call(
        "mi355_attn_bwd_qnorm", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
        L.ptr(do), do.stride(0), L.ptr(lse), L.ptr(delta), L.ptr(dk), dk.stride(0), L.ptr(dv), dv.stride(0), L.ptr(key_mask), int(causal) | _ATTN_ABLATE | ready, scale,
        L.ptr(ws), need, L.ptr(qkv), qkv.stride(0), L.ptr(qw), L.ptr(cos), L.ptr(sin), L.ptr(pos), L.ptr(rstd), H, L.ptr(dqkv), dqkv.stride(0), L.ptr(part),
        L.ptr(cs16),
    )
    # one partial row per workgroup: summed in two fixed-order levels (a single launch over thousands of rows runs on two workgroups)
    dw = torch.empty(D, dtype=F32, device=q.device)
    fold = next((f for f in (64, 32, 16, 8) if parts % f == 0 and parts // f >= 8), 1)
    if fold > 1:
        mid = torch.empty(fold * D, dtype=F32, device=q.device)
        L.call("mi355_reduce_rows_f32", parts // fold, fold * D, L.ptr(part), L.ptr(mid), L.DT_F32, 0)
        L.call("mi355_reduce_rows_f32", fold, D, L.ptr(mid), L.ptr(dw), L.DT_F32, 0)
    else:
        L.call("mi355_reduce_rows_f32", parts, D, L.ptr(part), L.ptr(dw), L.DT_F32, 0)
    return dw


def cross_entropy(logits2d, targets, want_grad, grad_scale=None, inplace=True):
    """Row CE with ignore_index=-100 on bf16 logits.  Returns (loss_rows fp32, dlogits or None).  grad_scale: fp32 [1] device."""
    L.require_gpu(logits2d, targets)
    _rowmajor(logits2d, "logits")
    if logits2d.dtype != BF16 or targets.dtype != torch.int64 or targets.numel() != logits2d.shape[0] or not targets.is_contiguous():
        raise ValueError("cross_entropy: logits bf16 [rows,V], targets contiguous int64 [rows]")
    rows, V = logits2d.shape
    loss_rows = torch.empty(rows, dtype=F32, device=logits2d.device)
    dl = None
    if want_grad:
        # the gradient shares the logits' row pitch (the kernel addresses both with one leading dimension)
        dl = logits2d if inplace else torch.empty_strided(tuple(logits2d.shape), logits2d.stride(), dtype=logits2d.dtype, device=logits2d.device)
    L.call("mi355_cross_entropy", rows, V, L.ptr(logits2d), logits2d.stride(0), L.ptr(targets), L.ptr(loss_rows), L.ptr(dl), L.ptr(grad_scale))
    return loss_rows, dl


def ce_finalize(loss_rows, targets):
    L.require_gpu(loss_rows, targets)
    out3 = torch.empty(3, dtype=F32, device=loss_rows.device)
    L.call("mi355_ce_finalize", loss_rows.numel(), L.ptr(loss_rows), L.ptr(targets), L.ptr(out3))
    return out3


def embedding_fwd(ids, table, out=None):
    L.require_gpu(ids, table)
    if ids.dtype != torch.int64 or table.dtype != BF16 or not table.is_contiguous():
        raise ValueError("embedding_fwd: ids int64, table contiguous bf16")
    flat = ids.reshape(-1).contiguous()
    width = table.shape[1]
    if out is None:
        out = torch.empty((flat.numel(), width), dtype=BF16, device=table.device)
    _rowmajor(out, "out")
    L.call("mi355_embedding_fwd", flat.numel(), width, table.shape[0], L.ptr(flat), L.ptr(table), L.ptr(out), out.stride(0))
    return out


def embedding_bwd(ids, dout, dtable_f32):
    L.require_gpu(ids, dout, dtable_f32)
    flat = ids.reshape(-1).contiguous()
    _rowmajor(dout, "dout")
    if dtable_f32.dtype != F32 or not dtable_f32.is_contiguous() or dout.shape[0] != flat.numel():
        raise ValueError("embedding_bwd: accumulator must be contiguous fp32 and dout must have one row per id")
    L.call("mi355_embedding_bwd", flat.numel(), dout.shape[1], dtable_f32.shape[0], L.ptr(flat), L.ptr(dout), dout.stride(0), L.ptr(dtable_f32))


def embedding_bwd_sorted(ids, dout, table, accumulate, scale=1.0):
    """table[id] = (accumulate ? table[id] : 0) + scale * sum of dout rows whose id it is -- deterministic (no atomics): the ids are sorted
    stably first (index preparation, one device-side sort), then each run of equal ids is summed in token order by a fixed two-level tree over
    blocks of 32 sorted positions (a run of thousands of padding / placeholder tokens does not serialise on one wave).
    Rows of ``table`` (bf16 [vocab, width], row-strided views allowed) that no id names are left untouched."""
    L.require_gpu(ids, dout, table)
    flat = ids.reshape(-1).contiguous()
    _rowmajor(dout, "dout")
    _rowmajor(table, "table")
    if dout.dtype != BF16 or table.dtype != BF16 or dout.shape[0] != flat.numel() or dout.shape[1] != table.shape[1] or flat.dtype != torch.int64:
        raise ValueError("embedding_bwd_sorted: int64 ids, bf16 dout [tokens, width] and bf16 table [vocab, width]")
    sid, perm = torch.sort(flat, stable=True)
    need = L.load().mi355_embedding_bwd_sorted_workspace_bytes(flat.numel(), dout.shape[1])
    ws = torch.empty((need + 3) // 4, dtype=F32, device=dout.device)  # per-block parts between the two passes of the fixed summation tree
    L.call("mi355_embedding_bwd_sorted", flat.numel(), dout.shape[1], table.shape[0], L.ptr(sid), L.ptr(perm), L.ptr(dout), dout.stride(0), float(scale),
           L.ptr(table), table.stride(0), int(bool(accumulate)), L.ptr(ws), need)
    return table


def transpose(x, out=None):
    """out[c, r] = x[r, c] for a bf16 matrix (row-strided views allowed; rows, cols and pitches multiples of 8)."""
    L.require_gpu(x, out)
    _rowmajor(x, "x")
    if x.dtype != BF16:
        raise TypeError("transpose: bf16 only")
    R, C = x.shape
    if out is None:
        out = torch.empty((C, R), dtype=BF16, device=x.device)
    _rowmajor(out, "out")
    if tuple(out.shape) != (C, R) or out.dtype != BF16:
        raise ValueError("transpose: output must be bf16 [cols, rows]")
    L.call("mi355_transpose_bf16", R, C, L.ptr(x), x.stride(0), L.ptr(out), out.stride(0))
    return out


def dgrad(dy, w, out=None):
    """dX = dY W for a Linear with weight W [N_out, N_in].  Above ``DGRAD_NT_MIN_ROWS`` token rows W is transposed first (one pass over the
    weight, 63 MB per transformer block) and the product runs in the K-contiguous NT form, whose main loop needs no transposing LDS reads;
    same-box A/B in DESIGN.md.  The result is the same sum in the same order per K-tile, bit-identical to the NN form."""
    M = dy.shape[0]
    if DGRAD_NT and M >= DGRAD_NT_MIN_ROWS and (w.shape[0] & 7) == 0 and (w.shape[1] & 7) == 0:
        return gemm(L.GEMM_NT, dy, transpose(w), out=out)
    return gemm(L.GEMM_NN, dy, w, out=out)


def copy2d(src, dst):
    """dst[r,:] = src[r,:] for 2-D views with unit inner stride (bit-exact)."""
    L.require_gpu(src, dst)
    _rowmajor(src, "src")
    _rowmajor(dst, "dst")
    if src.shape != dst.shape or src.dtype != dst.dtype:
        raise ValueError("copy2d: shape/dtype mismatch")
    es = src.element_size()
    L.call("mi355_copy2d", src.shape[0], src.shape[1] * es, L.ptr(src), src.stride(0) * es, L.ptr(dst), dst.stride(0) * es)
    return dst


def patchify(img, patch, out_dtype=BF16):
    L.require_gpu(img)
    if img.dtype != F32 or img.dim() != 4 or not img.is_contiguous():
        raise ValueError("patchify: image must be contiguous fp32 NCHW")
    B, C, H, W = img.shape
    rows = torch.empty((B * (H // patch) * (W // patch), C * patch * patch), dtype=out_dtype, device=img.device)
    L.call("mi355_patchify", B, C, H, W, patch, L.ptr(img), L.ptr(rows), L.dt_code(out_dtype))
    return rows


def layernorm_fwd(x2d, scale, shift, out_dtype=BF16, eps=1e-5, want_stats=False, mode=0):
    L.require_gpu(x2d, scale, shift)
    if x2d.dtype != F32 or not x2d.is_contiguous() or scale.dtype != F32 or shift.dtype != F32:
        raise ValueError("layernorm_fwd: x/scale/shift must be fp32, x contiguous")
    split = isinstance(out_dtype, str) and out_dtype == "split3"  # bf16 [rows, 3 * width] = [hi | lo | hi] of the fp32 result (see ``split3``)
    y = torch.empty((x2d.shape[0], 3 * x2d.shape[1]) if split else x2d.shape, dtype=BF16 if split else out_dtype, device=x2d.device)
    mean = rsig = None
    if want_stats:
        mean = torch.empty(x2d.shape[0], dtype=F32, device=x2d.device)
        rsig = torch.empty_like(mean)
    L.call("mi355_layernorm_fwd", x2d.shape[0], x2d.shape[1], L.ptr(x2d), L.ptr(scale), L.ptr(shift), L.ptr(y), L.DT_SPLIT3 if split else L.dt_code(out_dtype), L.ptr(mean), L.ptr(rsig), eps, mode)
    return (y, mean, rsig) if want_stats else y


def cast(src, dtype):
    L.require_gpu(src)
    if src.dtype == dtype:
        return src
    src = src.contiguous()
    dst = torch.empty(src.shape, dtype=dtype, device=src.device)
    L.call("mi355_cast", src.numel(), L.ptr(src), L.dt_code(src.dtype), L.ptr(dst), L.dt_code(dtype))
    return dst


def split3(x2d, weight_order=False):
    """fp32 [rows, K] (row-strided view allowed) -> bf16 [rows, 3K]: [hi | lo | hi] (activations) or [hi | hi | lo] (``weight_order``), hi = bf16(x),
    lo = bf16(x - hi).  One NT GEMM over K' = 3K of an activation block against a weight block is the fp32-grade product (csrc/tower_f32.hip)."""
    L.require_gpu(x2d)
    _rowmajor(x2d, "split3 input")
    if x2d.dtype != F32 or x2d.shape[1] % 8 or x2d.stride(0) % 4:
        raise ValueError("split3: fp32 rows with a width that is a multiple of 8 and a row pitch that is a multiple of 4")
    out = torch.empty((x2d.shape[0], 3 * x2d.shape[1]), dtype=BF16, device=x2d.device)
    L.call("mi355_split3_bf16", x2d.shape[0], x2d.shape[1], L.ptr(x2d), x2d.stride(0), L.ptr(out), int(bool(weight_order)))
    return out


def attn_f32_fwd(q, k, v, B, S, H, D, scale=None):
    """softmax(q k^T scale) v on fp32 [B*S, H*D] operands (row-strided views allowed), every key visible; returns fp32 [B*S, H*D]."""
    L.require_gpu(q, k, v)
    for t, n in ((q, "q"), (k, "k"), (v, "v")):
        _rowmajor(t, n)
        if t.dtype != F32 or tuple(t.shape) != (B * S, H * D) or t.stride(0) % 4:
            raise ValueError(f"attn_f32_fwd: {n} must be fp32 [{B * S}, {H * D}] with a row pitch that is a multiple of 4")
    o = torch.empty((B * S, H * D), dtype=F32, device=q.device)
    L.call("mi355_attn_f32_fwd", B, S, H, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0), D ** -0.5 if scale is None else scale)
    return o


def vit_embed_assemble(proj, cls, pos, B, S, width):
    L.require_gpu(proj, cls, pos)
    out = torch.empty((B, S, width), dtype=F32, device=proj.device)
    L.call("mi355_vit_embed_assemble", B, S, width, L.ptr(proj), L.ptr(cls), L.ptr(pos), L.ptr(out))
    return out


_SUMSQ_PARTS = {}


def _sumsq_parts(device):
    """Per (device, stream) scratch of mi355_sumsq's per-block partial sums (4096 floats): calls on different streams never share one."""
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _SUMSQ_PARTS.get(key)
    if buf is None:
        if len(_SUMSQ_PARTS) > 16:
            _SUMSQ_PARTS.clear()
        buf = _SUMSQ_PARTS[key] = torch.empty(4096, dtype=F32, device=device)
    return buf


def sumsq_into(x, acc):
    L.require_gpu(x, acc)
    x = x if x.is_contiguous() else x.contiguous()
    L.call("mi355_sumsq", x.numel(), L.ptr(x), L.dt_code(x.dtype), L.ptr(acc), L.ptr(_sumsq_parts(x.device)))


def clip_scale_(x, sumsq, max_norm):
    L.require_gpu(x, sumsq)
    if not x.is_contiguous():
        raise ValueError("clip_scale_: tensor must be contiguous")
    L.call("mi355_clip_scale", x.numel(), L.ptr(x), L.dt_code(x.dtype), L.ptr(sumsq), float(max_norm))


def adamw_(param_flat, grad_flat, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, sumsq=None, max_norm=0.0):
    """One AdamW step over a flat buffer (see include/mi355_vlm.h::mi355_adamw); sumsq = device scalar of the global squared
    gradient norm fuses clip_grad_norm_(max_norm) into the update."""
    L.require_gpu(param_flat, grad_flat, exp_avg, exp_avg_sq, sumsq)
    n = param_flat.numel()
    if not (param_flat.is_contiguous() and grad_flat.is_contiguous() and grad_flat.numel() == n):
        raise ValueError("adamw_: parameter and gradient must be contiguous buffers of one size")
    if exp_avg.dtype != F32 or exp_avg_sq.dtype != F32 or exp_avg.numel() != n or exp_avg_sq.numel() != n:
        raise ValueError("adamw_: moments must be fp32 buffers of the parameter's size")
    L.call("mi355_adamw", n, L.ptr(param_flat), L.dt_code(param_flat.dtype), L.ptr(grad_flat), L.dt_code(grad_flat.dtype), L.ptr(exp_avg),
           L.ptr(exp_avg_sq), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step), L.ptr(sumsq), float(max_norm))


def add_f32_to_bf16(a_f32, b_bf16, dst_bf16):
    L.require_gpu(a_f32, dst_bf16)
    L.call("mi355_add_f32_to_bf16", a_f32.numel(), L.ptr(a_f32), L.ptr(b_bf16), L.ptr(dst_bf16))


def scale_bf16(x, scale_f32, out=None):
    """out = x * scale (device fp32 scalar tensor)."""
    L.require_gpu(x, scale_f32)
    if x.dtype != BF16 or not x.is_contiguous() or scale_f32.dtype != F32:
        raise ValueError("scale_bf16: x contiguous bf16, scale fp32 device scalar")
    out = torch.empty_like(x) if out is None else out
    L.call("mi355_scale_bf16", x.numel(), L.ptr(x), L.ptr(scale_f32), L.ptr(out))
    return out


def gelu_fwd(x, tanh=False):
    L.require_gpu(x)
    if x.dtype != BF16 or not x.is_contiguous() or x.numel() % 8:
        raise ValueError("gelu_fwd: contiguous bf16 with numel % 8 == 0")
    y = torch.empty_like(x)
    L.call("mi355_gelu_fwd", x.numel(), L.ptr(x), L.ptr(y), 1 if tanh else 0)
    return y


def gelu_bwd(x, dy, tanh=False):
    L.require_gpu(x, dy)
    if dy.dtype != BF16 or not dy.is_contiguous() or dy.shape != x.shape:
        raise ValueError("gelu_bwd: dy must be contiguous bf16 like x")
    dx = torch.empty_like(x)
    L.call("mi355_gelu_bwd", x.numel(), L.ptr(x), L.ptr(dy), L.ptr(dx), 1 if tanh else 0)
    return dx


def layernorm_bwd(x2d, scale, mean, rsig, dy, dres=None, eps=1e-5, dscale_out=None, dshift_out=None, accumulate=False, mode=0):
    """Returns dx fp32 (+ dres).  dscale/dshift go to the given fp32 destinations (optionally accumulated) or new tensors."""
    L.require_gpu(x2d, dy, dres)
    rows, width = x2d.shape
    if x2d.dtype != F32 or not x2d.is_contiguous() or not dy.is_contiguous() or dy.shape != x2d.shape:
        raise ValueError("layernorm_bwd: x fp32 contiguous, dy contiguous of the same shape")
    if dres is not None and (dres.dtype != F32 or not dres.is_contiguous() or dres.shape != x2d.shape):
        raise ValueError("layernorm_bwd: dres must be contiguous fp32 like x")
    dx = torch.empty_like(x2d)
    parts = min(1024, (rows + 3) // 4)
    part = torch.empty((parts, 2 * width), dtype=F32, device=x2d.device)
    L.call("mi355_layernorm_bwd", rows, width, L.ptr(x2d), L.ptr(scale), L.ptr(mean), L.ptr(rsig), L.ptr(dy), L.dt_code(dy.dtype), L.ptr(dres), L.ptr(dx), L.ptr(part), parts, eps, mode)
    both = torch.empty(2 * width, dtype=F32, device=x2d.device)
    L.call("mi355_reduce_rows_f32", parts, 2 * width, L.ptr(part), L.ptr(both), L.DT_F32, 0)
    outs = []
    for src, dst in ((both[:width], dscale_out), (both[width:], dshift_out)):
        if dst is None:
            outs.append(src)
        else:
            if accumulate:
                dst.add_(src) if not dst.is_cuda else L.call("mi355_reduce_rows_f32", 1, width, L.ptr(src.contiguous()), L.ptr(dst), L.DT_F32, 1)
            else:
                L.call("mi355_reduce_rows_f32", 1, width, L.ptr(src.contiguous()), L.ptr(dst), L.DT_F32, 0)
            outs.append(dst)
    return dx, outs[0], outs[1]


def patchify3d(video, patch, tpatch, out_dtype=BF16):
    """video fp32 (B,C,T,H,W) -> rows [B*(T/tpatch)*gh*gw, C*tpatch*patch*patch] (Conv3d patch order)."""
    L.require_gpu(video)
    if video.dtype != F32 or video.dim() != 5 or not video.is_contiguous():
        raise ValueError("patchify3d: video must be contiguous fp32 (B,C,T,H,W)")
    B, C, T, H, W = video.shape
    rows = torch.empty((B * (T // tpatch) * (H // patch) * (W // patch), C * tpatch * patch * patch), dtype=out_dtype, device=video.device)
    L.call("mi355_patchify3d", B, C, T, H, W, patch, tpatch, L.ptr(video), L.ptr(rows), L.dt_code(out_dtype))
    return rows


def merge_patches(x2d, frames, gh, gw, m, inverse=False):
    """m x m spatial merge of patch rows [frames*gh*gw, d] -> [frames*(gh/m)*(gw/m), m*m*d] (or its inverse)."""
    L.require_gpu(x2d)
    if not x2d.is_contiguous():
        raise ValueError("merge_patches: contiguous rows required")
    es = x2d.element_size()
    if not inverse:
        if x2d.shape[0] != frames * gh * gw:
            raise ValueError("merge_patches: row count != frames*gh*gw")
        d = x2d.shape[1]
        out = torch.empty((frames * (gh // m) * (gw // m), m * m * d), dtype=x2d.dtype, device=x2d.device)
    else:
        d = x2d.shape[1] // (m * m)
        if x2d.shape[0] != frames * (gh // m) * (gw // m):
            raise ValueError("merge_patches(inverse): row count != frames*(gh/m)*(gw/m)")
        out = torch.empty((frames * gh * gw, d), dtype=x2d.dtype, device=x2d.device)
    L.call("mi355_merge_patches", frames, gh, gw, m, d * es, L.ptr(x2d), L.ptr(out), int(inverse))
    return out


def scatter_rows_fwd(emb2d, vis2d, mask_u8, slot_i32):
    L.require_gpu(emb2d, vis2d, mask_u8, slot_i32)
    if not (emb2d.is_contiguous() and vis2d.is_contiguous() and emb2d.dtype == vis2d.dtype and emb2d.shape[1] == vis2d.shape[1]):
        raise ValueError("scatter_rows_fwd: contiguous rows of equal width and dtype required")
    if not (mask_u8.dtype == torch.uint8 and slot_i32.dtype == torch.int32 and mask_u8.numel() == emb2d.shape[0] == slot_i32.numel()):
        raise ValueError("scatter_rows_fwd: mask uint8 [tokens], slot int32 [tokens]")
    out = torch.empty_like(emb2d)
    L.call("mi355_scatter_rows", emb2d.shape[0], emb2d.shape[1] * emb2d.element_size(), L.ptr(mask_u8), L.ptr(slot_i32), L.ptr(emb2d), L.ptr(vis2d), L.ptr(out), None, 0)
    return out


def scatter_rows_bwd(g2d, n_vis_rows, mask_u8, slot_i32):
    L.require_gpu(g2d, mask_u8, slot_i32)
    g2d = g2d if g2d.is_contiguous() else g2d.contiguous()
    d_emb = torch.empty_like(g2d)
    d_vis = torch.zeros((n_vis_rows, g2d.shape[1]), dtype=g2d.dtype, device=g2d.device)
    L.call("mi355_scatter_rows", g2d.shape[0], g2d.shape[1] * g2d.element_size(), L.ptr(mask_u8), L.ptr(slot_i32), L.ptr(g2d), None, L.ptr(d_emb), L.ptr(d_vis), 1)
    return d_emb, d_vis


# ----------------------------------------------------------------------------------------------- stand-alone RoPE, dropout
def _bhsd_strides(t, name):
    if t.dim() != 4 or t.stride(3) != 1:
        raise ValueError(f"{name}: expected (b, heads, seq, head_dim) with unit inner stride, got shape {tuple(t.shape)} strides {t.stride()}")
    return t.stride(0), t.stride(1), t.stride(2)


def rope_apply(x, cos_t, sin_t, idx=None, transpose=False):
    """RoPE on x (b, heads, s, head_dim) bf16 / fp32, any outer strides: rows of cos_t / sin_t (fp32 [rows, R], R <= head_dim) are
    picked by idx (int32 [b*s]) or by the sequence position.  transpose=True applies the adjoint (backward).  The result has x's layout."""
    L.require_gpu(x, cos_t, sin_t, idx)
    if x.dtype not in (BF16, F32):
        raise TypeError(f"rope_apply: bf16 or fp32 input expected, got {x.dtype}")
    sb, sh, ss = _bhsd_strides(x, "rope_apply")
    B, H, S, D = x.shape
    if cos_t.dtype != F32 or sin_t.dtype != F32 or cos_t.dim() != 2 or cos_t.shape != sin_t.shape or not cos_t.is_contiguous() or not sin_t.is_contiguous():
        raise ValueError("rope_apply: cos/sin must be contiguous fp32 [rows, R]")
    R = cos_t.shape[1]
    if R > D or R % 2:
        raise ValueError(f"rope_apply: rotation width {R} must be even and <= head_dim {D}")
    if idx is not None:
        if idx.dtype != torch.int32 or idx.numel() != B * S or not idx.is_contiguous():
            raise ValueError("rope_apply: idx must be contiguous int32 [b*s]")
    elif cos_t.shape[0] < S:
        raise ValueError(f"rope_apply: sequence length {S} exceeds the {cos_t.shape[0]} rows of the coefficient table")
    out = torch.empty_like(x)  # preserve_format: x's strides when x is dense (a transposed token-major view stays one), else contiguous
    ob, oh, os_ = _bhsd_strides(out, "rope_apply(out)")
    L.call("mi355_rope_apply", B, H, S, D, R, L.ptr(x), L.dt_code(x.dtype), sb, sh, ss, L.ptr(cos_t), L.ptr(sin_t), cos_t.shape[0], L.ptr(idx),
           L.ptr(out), ob, oh, os_, int(transpose))
    return out


def dropout(x, p, seed, offset, residual=None, out_dtype=None):
    """y = residual + keep ? x / (1 - p) : 0 with the Philox mask of (seed, offset); x contiguous bf16 / fp32; the backward is the
    same call on the incoming gradient."""
    L.require_gpu(x, residual)
    if not x.is_contiguous() or x.dtype not in (BF16, F32):
        raise ValueError("dropout: contiguous bf16 / fp32 input expected")
    out_dtype = x.dtype if out_dtype is None else out_dtype
    if residual is not None and (residual.dtype != out_dtype or residual.shape != x.shape or not residual.is_contiguous()):
        raise ValueError("dropout: residual must be contiguous, shaped like x, in the output dtype")
    if not 0.0 <= p < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    L.call("mi355_dropout", x.numel(), L.ptr(x), L.dt_code(x.dtype), L.ptr(residual), L.ptr(y), L.dt_code(out_dtype), float(p), int(seed), int(offset))
    return y


def attn_dropout_fwd(q, k, v, B, S, Hq, Hkv, D, p, seed, offset, causal=False, scale=None):
    """Softmax attention with dropout(p) on the normalised weights (ViT train mode).  Layout as attn_fwd."""
    L.require_gpu(q, k, v)
    _check_attn_operand(q, "q", B * S, Hq * D)
    _check_attn_operand(k, "k", B * S, Hkv * D)
    _check_attn_operand(v, "v", B * S, Hkv * D)
    o = torch.empty((B * S, Hq * D), dtype=BF16, device=q.device)
    lse = torch.empty((B, Hq, S), dtype=F32, device=q.device)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_dropout_fwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
           L.ptr(lse), int(causal), scale, float(p), int(seed), int(offset))
    return o, lse


def attn_dropout_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, p, seed, offset, causal=False, scale=None):
    L.require_gpu(q, k, v, o, do, lse, dq, dk, dv)
    for t, n, w in ((q, "q", Hq), (k, "k", Hkv), (v, "v", Hkv), (o, "o", Hq), (do, "do", Hq), (dq, "dq", Hq), (dk, "dk", Hkv), (dv, "dv", Hkv)):
        _check_attn_operand(t, n, B * S, w * D)
    if not (lse.dtype == F32 and lse.is_contiguous() and tuple(lse.shape) == (B, Hq, S)):
        raise ValueError("attention: lse must be contiguous fp32 [B,Hq,S]")
    delta = torch.empty_like(lse)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_dropout_bwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
           L.ptr(do), do.stride(0), L.ptr(lse), L.ptr(delta), L.ptr(dq), dq.stride(0), L.ptr(dk), dk.stride(0), L.ptr(dv), dv.stride(0),
           int(causal), scale, float(p), int(seed), int(offset))


def attn_generic_dropout_fwd(q, k, v, B, S, Hq, Hkv, D, p, seed, offset, key_mask=None, scale=None):
    """The SDPA call of (MRoPE)GatedAttention with ``dropout_p`` AND a padding mask: mask semantics of ``kernels_q35.attn_generic_fwd`` (causal, padded keys visible
    -- upstream's quirk), Philox dropout on the normalised weights as ``attn_dropout_fwd``.  Layout as attn_fwd."""
    L.require_gpu(q, k, v, key_mask)
    _check_attn_operand(q, "q", B * S, Hq * D)
    _check_attn_operand(k, "k", B * S, Hkv * D)
    _check_attn_operand(v, "v", B * S, Hkv * D)
    if key_mask is not None and not (key_mask.dtype == torch.uint8 and key_mask.is_contiguous() and tuple(key_mask.shape) == (B, S)):
        raise ValueError("attention: key_mask must be contiguous uint8 [B,S]")
    o = torch.empty((B * S, Hq * D), dtype=BF16, device=q.device)
    lse = torch.empty((B, Hq, S), dtype=F32, device=q.device)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_generic_dropout_fwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
           L.ptr(lse), L.ptr(key_mask), scale, float(p), int(seed), int(offset))
    return o, lse


def attn_generic_dropout_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, p, seed, offset, key_mask=None, scale=None):
    L.require_gpu(q, k, v, o, do, lse, dq, dk, dv, key_mask)
    for t, n, w in ((q, "q", Hq), (k, "k", Hkv), (v, "v", Hkv), (o, "o", Hq), (do, "do", Hq), (dq, "dq", Hq), (dk, "dk", Hkv), (dv, "dv", Hkv)):
        _check_attn_operand(t, n, B * S, w * D)
    if not (lse.dtype == F32 and lse.is_contiguous() and tuple(lse.shape) == (B, Hq, S)):
        raise ValueError("attention: lse must be contiguous fp32 [B,Hq,S]")
    delta = torch.empty_like(lse)
    scale = D ** -0.5 if scale is None else scale
    L.call("mi355_attn_generic_dropout_bwd", B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0),
           L.ptr(do), do.stride(0), L.ptr(lse), L.ptr(delta), L.ptr(dq), dq.stride(0), L.ptr(dk), dk.stride(0), L.ptr(dv), dv.stride(0),
           L.ptr(key_mask), scale, float(p), int(seed), int(offset))


def mfma_pipe_rate(seconds=2.0, blocks=256, shape="32x32x16"):
    """TFLOP/s the matrix pipe alone sustains on this board (``mi355_mfma_pipe_probe``: random bf16 operands in registers, no memory traffic), timed with HIP events over the
    second half of ``seconds`` of back-to-back launches -- the board's power cap needs about a second to settle the clock.  A measurement aid for ``bench.py``."""
    import torch

    out = torch.empty(blocks * 256, dtype=torch.float32, device="cuda")
    L.require_gpu(out)
    reps = 8000  # ~ 1 ms a launch
    flop = 2.0 * 32 * 32 * 16 * 16 * reps * 4 * blocks
    if shape == "16x16x32":  # the shape the GEMMs issue: 32 products per repetition (the entry point takes -reps for it)
        flop = 2.0 * 16 * 16 * 32 * 32 * reps * 4 * blocks
        reps = -reps
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

    def run(n):
        ev[0].record()
        for _ in range(n):
            L.call("mi355_mfma_pipe_probe", blocks, reps, L.ptr(out))
        ev[1].record()
        torch.cuda.synchronize()
        return ev[0].elapsed_time(ev[1]) / n  # ms per launch

    per = run(20)
    n = max(20, int(seconds * 500 / per))
    run(n)  # first half: the clock settles
    per = run(n)
    return flop / (per * 1e-3) / 1e12
