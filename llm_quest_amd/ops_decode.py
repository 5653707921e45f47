"""Inference with a KV cache through Qwen3: ``Qwen3Model.forward(x, kv_cache=...)`` as
``generate_loop_kv_cache`` drives it (llm_quest/generate.py:97-151, qwen3_attention.py:117-146).

Two shapes of call, both without autograd:
  * prefill -- the whole prompt, cache empty: the training-path kernels (MFMA GEMMs, flash attention over the prompt), plus a
    strided copy of each layer's post-norm / post-RoPE K rows and V rows into the token-major cache;
  * decode  -- one new token per sequence: every linear layer is a weight stream (``mi355_gemv_bf16``, <= 8 sequences; larger
    batches fall back to the GEMM), attention is one query row per head over the cache (``mi355_attn_decode``).
"""

import torch

from . import _lib as L
from . import kernels as K
from . import ops

BF16 = torch.bfloat16


def gemv(x2d, w, residual=None):
    """x [M <= 8, K] @ w[N, K]^T (+ residual); larger M goes through the MFMA GEMM."""
    M = x2d.shape[0]
    if M > 8:
        return K.gemm(L.GEMM_NT, x2d, w, residual=residual)
    L.require_gpu(x2d, w, residual)
    if x2d.dtype != BF16 or w.dtype != BF16 or x2d.stride(1) != 1 or w.stride(1) != 1 or x2d.shape[1] != w.shape[1]:
        raise ValueError("gemv: bf16 operands with unit inner stride and a common K")
    y = torch.empty((M, w.shape[0]), dtype=BF16, device=x2d.device)
    L.call("mi355_gemv_bf16", M, w.shape[0], w.shape[1], L.ptr(x2d), x2d.stride(0), L.ptr(w), w.stride(0), L.ptr(y), y.stride(0),
           L.ptr(residual), residual.stride(0) if residual is not None else 0)
    return y


def gemv_pro(x2d, w, prologue, norm_w=None, eps=1e-6, residual=None):
    """The decode step's weight stream with the row operation in front of it folded in (``mi355_gemv_bf16_pro``): prologue "rmsnorm" = ``gemv(rmsnorm_fwd(x, norm_w), w)``,
    "swiglu" = ``gemv(swiglu_fwd(x, K), w)`` with x = fused lin1 | lin_gate rows [M, 2K]; bit-identical to the two launches, one launch fewer each."""
    M = x2d.shape[0]
    Kd = w.shape[1]
    L.require_gpu(x2d, w, norm_w, residual)
    code = {"rmsnorm": 1, "swiglu": 2}[prologue]
    if M > 8 or x2d.dtype != BF16 or w.dtype != BF16 or x2d.stride(1) != 1 or w.stride(1) != 1 or x2d.shape[1] != (2 * Kd if code == 2 else Kd):
        raise ValueError("gemv_pro: <= 8 bf16 rows with unit inner stride, K (rmsnorm) or 2K (swiglu) wide")
    if code == 1 and (norm_w is None or norm_w.dtype != BF16 or norm_w.numel() != Kd or not norm_w.is_contiguous()):
        raise ValueError("gemv_pro: rmsnorm prologue needs a contiguous bf16 weight of K elements")
    y = torch.empty((M, w.shape[0]), dtype=BF16, device=x2d.device)
    L.call("mi355_gemv_bf16_pro", M, w.shape[0], Kd, L.ptr(x2d), x2d.stride(0), code, L.ptr(norm_w), eps, L.ptr(w), w.stride(0), L.ptr(y), y.stride(0),
           L.ptr(residual), residual.stride(0) if residual is not None else 0)
    return y


def fused_rows(M, Kd):
    """Whether a one-token step of M sequences takes the fused streams (the operand rows must fit the kernel's 64-KiB LDS image)."""
    return M <= 8 and M * Kd * 2 + 32 <= 65536  # (csrc/decode.hip: the operand image plus 32 bytes of row sums within 64 KiB)


def decode_splits(n_keys, B, Hq):
    """Workgroups per (sequence, head) of the decode attention: one streams about 256 keys per memory round trip (3.4 us), so longer caches are dealt to several -- their
    partial results cost a second, tiny launch (measured at batch 1: 1 868 keys 1.65 -> 1.13 ms per token) -- as many as keep the chip's 256 CUs busy once."""
    if n_keys <= 320:
        return 1
    return max(1, min((n_keys + 255) // 256, 16, max(1, 256 // (B * Hq))))


def _split_ws(splits, B, Hq, D, device):
    return torch.empty(B * Hq * splits * (D + 2), dtype=torch.float32, device=device) if splits > 1 else None


def attn_decode_qkv(qkv, qw, kw, cos, sin, pos, Hq, Hkv, D, kc, vc, write_pos_dev, len_dev, key_mask=None, scale=None, eps=1e-6, splits=1):
    """One new token per sequence from the fused QKV stream's raw rows to the attention context (``mi355_attn_decode_qkv``): QK-norm + RoPE of the query and the new key
    inside the launch, key and value heads written into cache row ``*write_pos_dev``, attention over ``*len_dev`` keys -- ``qknorm_rope_fwd`` + ``kv_append_dev`` +
    ``attn_decode`` in one launch, bit-identical."""
    L.require_gpu(qkv, qw, kw, cos, sin, pos, kc, vc, write_pos_dev, len_dev, key_mask)
    B = qkv.shape[0]
    if (qkv.dtype != BF16 or qkv.stride(1) != 1 or qkv.shape[1] != (Hq + 2 * Hkv) * D or kc.shape != vc.shape or kc.stride() != vc.stride() or kc.shape[0] != B
            or kc.shape[2] != Hkv * D or kc.stride(2) != 1 or pos.dtype != torch.int32 or pos.numel() != B or write_pos_dev.dtype != torch.int32 or len_dev.dtype != torch.int32):
        raise ValueError("attn_decode_qkv: qkv bf16 [B, (Hq + 2 Hkv) D], caches [B, capacity, Hkv D] of one layout, int32 positions / length on the device")
    ldm = 0
    if key_mask is not None:
        if key_mask.dtype != torch.uint8 or key_mask.shape[0] != B or key_mask.shape[1] < kc.shape[1] or key_mask.stride(1) != 1:
            raise ValueError("attn_decode_qkv: key_mask uint8 [B, >= capacity]")
        ldm = key_mask.stride(0)
    o = torch.empty((B, Hq * D), dtype=BF16, device=qkv.device)
    ws = _split_ws(splits, B, Hq, D, qkv.device)
    L.call("mi355_attn_decode_qkv", B, Hq, Hkv, D, L.ptr(qkv), qkv.stride(0), L.ptr(qw), L.ptr(kw), L.ptr(cos), L.ptr(sin), L.ptr(pos), L.ptr(kc), L.ptr(vc), kc.stride(0),
           kc.stride(1), kc.shape[1], L.ptr(write_pos_dev), L.ptr(len_dev), L.ptr(key_mask), ldm, L.ptr(o), D ** -0.5 if scale is None else scale, eps, splits, L.ptr(ws))
    return o


def decode_advance(next_ids, tok, rope_pos, write_pos, length):
    """tok <- next_ids; every device-side counter of the step + 1 (one launch: the tail of ``GraphDecoder``'s captured step)."""
    L.require_gpu(next_ids, tok, rope_pos, write_pos, length)
    B = tok.numel()
    if next_ids.dtype != torch.int64 or tok.dtype != torch.int64 or next_ids.numel() != B or rope_pos.numel() != B or not (next_ids.is_contiguous() and tok.is_contiguous()):
        raise ValueError("decode_advance: int64 ids of one size, int32 counters")
    L.call("mi355_decode_advance", B, L.ptr(next_ids), L.ptr(tok), L.ptr(rope_pos), L.ptr(write_pos), L.ptr(length))


def attn_decode(q, kc, vc, length, Hq, Hkv, D, key_mask=None, scale=None, len_dev=None, splits=None):
    """q [B, Hq*D]; kc / vc [B, capacity, Hkv*D]; attends to keys [0, length) -- or [0, min(length, *len_dev)) with the length read
    on the device (graph replay)."""
    L.require_gpu(q, kc, vc, key_mask)
    B = q.shape[0]
    if not (q.is_contiguous() and q.dtype == BF16 and kc.dtype == BF16 and kc.shape == vc.shape and kc.stride() == vc.stride() and kc.stride(2) == 1):
        raise ValueError("attn_decode: contiguous bf16 q, caches [B, capacity, Hkv*D] of one layout")
    if length > kc.shape[1]:
        raise ValueError("attn_decode: length exceeds the cache capacity")
    ldm = 0
    if key_mask is not None:
        if key_mask.dtype != torch.uint8 or key_mask.shape[0] != B or key_mask.shape[1] < length or key_mask.stride(1) != 1:
            raise ValueError("attn_decode: key_mask uint8 [B, >= length]")
        ldm = key_mask.stride(0)
    o = torch.empty_like(q)
    if splits is None:
        splits = decode_splits(length, B, Hq) if len_dev is None else 1  # with the length on the device the caller knows how long the cache will get
    ws = _split_ws(splits, B, Hq, D, q.device)
    L.call("mi355_attn_decode", B, Hq, Hkv, D, L.ptr(q), L.ptr(kc), L.ptr(vc), kc.stride(0), kc.stride(1), length, L.ptr(len_dev), L.ptr(key_mask), ldm, L.ptr(o),
           D ** -0.5 if scale is None else scale, splits, L.ptr(ws))
    return o


def argmax_rows(logits2d):
    L.require_gpu(logits2d)
    if logits2d.dtype != BF16 or logits2d.dim() != 2 or logits2d.stride(1) != 1:
        raise ValueError("argmax_rows: bf16 [rows, V] with unit inner stride")
    out = torch.empty(logits2d.shape[0], dtype=torch.int64, device=logits2d.device)
    ws = torch.empty(logits2d.shape[0] * 64 * 12 // 8 + 2, dtype=torch.int64, device=logits2d.device)
    L.call("mi355_argmax_rows", logits2d.shape[0], logits2d.shape[1], L.ptr(logits2d), logits2d.stride(0), L.ptr(out), L.ptr(ws))
    return out


def kv_append_dev(k_rows, v_rows, kc, vc, write_pos_dev):
    """cache[b, *write_pos, :] = rows[b, :] (one decoded token per sequence, position on the device)."""
    L.require_gpu(k_rows, v_rows, kc, vc, write_pos_dev)
    B, width = k_rows.shape
    if kc.shape != vc.shape or kc.stride() != vc.stride() or kc.shape[0] != B or kc.shape[2] != width or write_pos_dev.dtype != torch.int32:
        raise ValueError("kv_append_dev: caches [B, capacity, width] of one layout, position int32 on the device")
    L.call("mi355_kv_append", B, width, L.ptr(k_rows), k_rows.stride(0), L.ptr(v_rows), v_rows.stride(0), L.ptr(kc), L.ptr(vc), kc.stride(0), kc.stride(1),
           kc.shape[1], L.ptr(write_pos_dev))


def _lin(decode):
    return gemv if decode else (lambda a, w, residual=None: K.gemm(L.GEMM_NT, a, w, residual=residual))


def cached_positions(kv_cache, B, S, device, position_ids=None, reference_default=False):
    """int32 [B*S] rotary positions of the new tokens.  ``reference_default``: with ``position_ids=None`` RoPE.apply rotates by
    0..S-1 whatever the cache holds (common/rope.py:229-232) -- the stand-alone modules keep that; ``Qwen3Model`` continues at ``start_pos``."""
    if position_ids is not None:
        return position_ids.to(device=device, dtype=torch.int32).expand(B, S).reshape(-1).contiguous()
    start = 0 if reference_default else kv_cache.start_pos
    return (start + torch.arange(S, dtype=torch.int32, device=device)).repeat(B)


def cached_key_mask(attn_mask, kv_cache, B, S, device):
    if attn_mask is None:
        return None
    km = attn_mask.to(device=device, dtype=torch.uint8).contiguous()
    if km.shape[0] != B or km.shape[1] < kv_cache.start_pos + S:
        raise ValueError(f"attn_mask must cover the cached sequence: (b, >= {kv_cache.start_pos + S}), got {tuple(km.shape)}")
    return km


@torch.no_grad()
def attention_cached(att, h1, B, S, cos, sin, pos, km, kv_cache, dev_state=None, residual=None, pre_norm=None):
    """``GroupedQueryAttention.forward(..., kv_cache=...)`` (reference qwen3_attention.py:91-148) on rows h1 bf16 [B*S, d_in]: fused QKV
    projection, QK-norm + RoPE, cache append, attention over the cache (prefill: flash attention over the prompt; decode: one query row per
    head), output projection (+ residual).  ``pre_norm``: h1 is the block's residual stream and this RMSNorm weight is applied inside the QKV
    stream (one-token steps, ``gemv_pro``)."""
    arena = ops.arena_for(att)
    Hq, Hkv, D = att.num_heads, att.num_kv_groups, att.head_dim
    decode = kv_cache.start_pos > 0
    if decode and S != 1:
        raise NotImplementedError("with a filled KV cache one new token per sequence is decoded (q_seq_len 1, generate.py:139-148)")
    lin = _lin(decode)
    wqkv = arena.fused(att.w_queries.weight, att.w_values.weight)
    qkv = gemv_pro(h1, wqkv, "rmsnorm", pre_norm) if pre_norm is not None else lin(h1, wqkv)
    if dev_state is not None and D in (64, 128) and (km is None or km.shape[1] >= kv_cache.keys_cache[att.layer_idx].shape[1]):
        # graph replay, one launch from the raw QKV rows to the context: the new key / value go into the cache row at the device-side position
        write_pos, len_dev = dev_state[1], dev_state[2]
        splits = dev_state[3] if len(dev_state) > 3 else 1
        kc, vc = kv_cache.keys_cache[att.layer_idx], kv_cache.values_cache[att.layer_idx]
        ctx = attn_decode_qkv(qkv, att.q_norm.weight, att.k_norm.weight, cos, sin, pos, Hq, Hkv, D, kc, vc, write_pos, len_dev, key_mask=km, scale=att.att_scaling, splits=splits)
        return lin(ctx, att.out_proj.weight, residual=residual)
    q, k, _ = K.qknorm_rope_fwd(qkv, att.q_norm.weight, att.k_norm.weight, cos, sin, pos, Hq, Hkv, D)
    v = qkv[:, (Hq + Hkv) * D :]
    if dev_state is not None:  # graph replay: position and length are read on the device, the cache object is advanced by the caller
        write_pos, len_dev = dev_state[1], dev_state[2]
        kc, vc = kv_cache.keys_cache[att.layer_idx], kv_cache.values_cache[att.layer_idx]
        kv_append_dev(k, v, kc, vc, write_pos)
        ctx = attn_decode(q, kc, vc, kc.shape[1], Hq, Hkv, D, key_mask=km, scale=att.att_scaling, len_dev=len_dev)
    elif decode:
        kc, vc, end = kv_cache.append_rows(k, v, att.layer_idx, B, S)
        ctx = attn_decode(q, kc, vc, end, Hq, Hkv, D, key_mask=km, scale=att.att_scaling)
    else:
        kv_cache.append_rows(k, v, att.layer_idx, B, S)
        ctx, _ = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=None if km is None else km[:, :S].contiguous(), causal=True, scale=att.att_scaling)
    return lin(ctx, att.out_proj.weight, residual=residual)


@torch.no_grad()
def block_cached(blk, h, B, S, cos, sin, pos, km, kv_cache, dev_state=None):
    """``TransformerBlock.forward(..., kv_cache=...)`` (reference qwen3_transformer_block.py:91-112) on the bf16 residual rows h [B*S, d]."""
    arena = ops.arena_for(blk)
    ffn = blk.ffn
    decode = kv_cache.start_pos > 0
    lin = _lin(decode)
    F = ffn.lin1.weight.shape[0]
    if decode and fused_rows(B * S, max(h.shape[1], F)):  # one-token step: 6 launches instead of 10 (the norms and the activation ride in the weight streams)
        h = attention_cached(blk.att, h, B, S, cos, sin, pos, km, kv_cache, dev_state, residual=h, pre_norm=blk.norm1.weight)
        gu = gemv_pro(h, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight), "rmsnorm", blk.norm2.weight)
        return gemv_pro(gu, ffn.lin2.weight, "swiglu", residual=h)
    h1, _ = K.rmsnorm_fwd(h, blk.norm1.weight, want_rstd=False)
    h = attention_cached(blk.att, h1, B, S, cos, sin, pos, km, kv_cache, dev_state, residual=h)
    h2, _ = K.rmsnorm_fwd(h, blk.norm2.weight, want_rstd=False)
    gu = lin(h2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
    return lin(K.swiglu_fwd(gu, F), ffn.lin2.weight, residual=h)



@torch.no_grad()
def qwen3_forward_cached(model, x, kv_cache, attn_mask=None, position_ids=None, input_embedded=False, dev_state=None):
    """Logits (b, s, vocab) of ``Qwen3Model`` with ``kv_cache`` updated in place (reference: qwen3_model.py:60-94).

    ``dev_state = (rope_pos int32 [b], write_pos int32 [1], length int32 [1])`` (all on the device) switches the one-token step to
    device-side bookkeeping -- nothing the kernels are launched with changes from token to token, so the step can be captured in a
    hipGraph and replayed (``GraphDecoder``); the cache object's host counters are then advanced by the caller."""
    model._build_arenas()
    L.require_gpu(x)
    emb_w = model.emb_dict.weight
    if input_embedded:
        B, S, d = x.shape
        h = x.reshape(B * S, d).contiguous()
    else:
        B, S = x.shape
        h = K.embedding_fwd(x, emb_w)
    start = kv_cache.start_pos
    decode = start > 0
    if decode and S != 1:
        raise NotImplementedError("with a filled KV cache one new token per sequence is decoded (q_seq_len 1, generate.py:139-148)")
    if dev_state is not None:
        if not decode:
            raise ValueError("device-side bookkeeping applies to one-token decode steps only")
        pos, write_pos, len_dev = dev_state[:3]
    elif position_ids is not None:
        pos = position_ids.to(device=h.device, dtype=torch.int32).expand(B, S).reshape(-1).contiguous()
    else:
        pos = (start + torch.arange(S, dtype=torch.int32, device=h.device)).repeat(B)
    km = None
    if attn_mask is not None:
        km = attn_mask.to(device=h.device, dtype=torch.uint8).contiguous()
        if km.shape[0] != B or km.shape[1] < start + S:
            raise ValueError(f"attn_mask must cover the cached sequence: (b, >= {start + S}), got {tuple(km.shape)}")
    for blk in model.trf_blocks:
        h = block_cached(blk, h, B, S, model.cos, model.sin, pos, km, kv_cache, dev_state)
    if decode and fused_rows(B * S, h.shape[1]):
        return gemv_pro(h, model.out_head.weight, "rmsnorm", model.final_norm.weight).view(B, S, -1)
    lin = _lin(decode)
    hn, _ = K.rmsnorm_fwd(h, model.final_norm.weight, want_rstd=False)
    return lin(hn, model.out_head.weight).view(B, S, -1)


class GraphDecoder:
    """Greedy one-token decode steps of ``Qwen3Model`` as ONE captured hipGraph, replayed per token.

    A decode step is ~340 small launches (28 layers x 12 kernels); launched one by one from Python the step is launch-bound
    (3.9 ms at batch 1 for 2.4 ms of kernels).  Everything that changes from token to token -- the token ids, the rotary position,
    the cache write position and length -- lives in device buffers that the graph's own tail updates, so replay needs no host work.
    The cache must already hold the prompt (prefill) and have room for every token to come (``reserve``): its buffers may not move.
    """

    def __init__(self, model, kv_cache, first_token, max_new_tokens):
        L.require_gpu(first_token)
        self.model, self.kv = model, kv_cache
        B = first_token.shape[0]
        start = kv_cache.start_pos
        need = start + max_new_tokens
        if need > kv_cache.context_len:
            raise ValueError(f"GraphDecoder: {need} tokens exceed context_len {kv_cache.context_len}")
        kv_cache.reserve(need)
        dev = first_token.device
        self.tok = first_token.reshape(B, 1).to(torch.int64).clone()
        self.rope_pos = torch.full((B,), start, dtype=torch.int32, device=dev)
        self.write_pos = torch.tensor([start], dtype=torch.int32, device=dev)
        self.length = torch.tensor([start + 1], dtype=torch.int32, device=dev)
        self.splits = decode_splits(need, B, model.trf_blocks[0].att.num_heads)  # fixed for the captured graph: sized for the longest cache it will see
        self.capacity = need  # the reserved cache rows: the kernels neither write nor attend a token beyond them, so step() refuses to go there
        self.graph = None

    def _step(self):
        logits = qwen3_forward_cached(self.model, self.tok, self.kv, dev_state=(self.rope_pos, self.write_pos, self.length, self.splits))
        nxt = argmax_rows(logits.view(logits.shape[0], -1))
        decode_advance(nxt, self.tok, self.rope_pos, self.write_pos, self.length)

    def step(self):
        """Consumes ``self.tok`` (the token chosen last), leaves the next greedy token in ``self.tok`` and returns a copy of it."""
        if self.kv.start_pos >= self.capacity:
            raise RuntimeError(f"GraphDecoder: the reserved cache holds {self.capacity} tokens (prompt + max_new_tokens) and all of them are used: a further step "
                               "would neither store nor attend the current token")
        if self.graph is None:
            self._step()  # first step eagerly: loads every kernel before capture
            self._advance_host()
            out = self.tok.clone()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._step()
            self.graph = g
            _live.add(self)
            # capture does not execute: state is still the one after the eager step
            return out
        self.graph.replay()
        self._advance_host()
        return self.tok.clone()

    def _advance_host(self):
        self.kv.start_pos += 1
        self.kv.end_pos = self.kv.start_pos

    def close(self):
        """Destroy the captured graph NOW.  A hipGraph that is still alive when the interpreter shuts down is torn down after the
        HIP runtime's own teardown has begun and costs about a minute on ROCm 7.2 (measured); released here it takes 60 ms."""
        if self.graph is not None:
            torch.cuda.synchronize()
            self.graph = None
        _live.discard(self)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_live = set()


def _close_all():
    for d in list(_live):
        d.close()


import atexit  # noqa: E402

atexit.register(_close_all)


# ----------------------------------------------------------------------------------------------- Qwen3.5 hybrid stack with Qwen3_5Cache
class Q35Step:
    """What one cached step of the Qwen3.5 stack shares between its layers: shape, rotary rows, key mask, the cache, and whether the step decodes one
    token from carried state (GEMV projections, conv / recurrence updates, decode attention) or prefills."""

    __slots__ = ("B", "S", "cos_t", "sin_t", "pos", "km", "cache", "decode")

    def __init__(self, cache, B, S, device, cos, sin, mrope_section, attn_mask=None, position_ids=None, decode=None):
        from . import kernels_q35 as Q

        self.B, self.S, self.cache = B, S, cache
        # a layer on its own: one new token against carried state (conv / recurrent state of some linear layer, or cached K / V rows) decodes
        self.decode = (S == 1 and (cache.has_previous_state or cache.kv_cache.start_pos > 0)) if decode is None else decode
        start = cache.kv_cache.start_pos
        if position_ids is None:  # text-only: 1-D rotary rows start .. start+S-1 of the model's table
            self.cos_t, self.sin_t = cos, sin
            self.pos = (start + torch.arange(S, dtype=torch.int32, device=device)).repeat(B)
        else:
            self.cos_t, self.sin_t = Q.mrope_table(cos, sin, position_ids.to(device), mrope_section)
            self.pos = torch.arange(B * S, dtype=torch.int32, device=device)
        self.km = None if attn_mask is None else attn_mask.to(device=device, dtype=torch.uint8).contiguous()

    def lin(self, a, w, residual=None):
        return gemv(a, w, residual=residual) if self.decode else K.gemm(L.GEMM_NT, a, w, residual=residual)


@torch.no_grad()
def q35_mixer_cached(att, arena, h1, st, i):
    """The token mixer of layer ``i`` on normalised rows ``h1`` [B*S, d] with the hybrid cache: FusedGatedDeltaNet (conv state + recurrent state,
    reference qwen3_5_text_model.py:96-191) or MRoPEGatedAttention (K / V rows, :206-267).  Returns the rows in front of ``out_proj``."""
    from . import kernels_q35 as Q
    from . import ops_q35

    B, S, cache, decode = st.B, st.S, st.cache, st.decode
    if att.is_linear:
        Hqk, Hv, Dk, Dv, QK, VG, C = ops_q35._gdn_dims(att)
        if st.km is not None:
            h1 = Q.rowmask(h1, st.km[:, -S:].reshape(-1).contiguous())
        proj = st.lin(h1, arena.fused(att.w_qkv.weight, att.w_alpha.weight))
        if decode:
            y = Q.causal_conv_silu_step(proj[:, :C], cache.get_conv_state(i), att.conv1d.weight)
        else:
            ks = att.conv_kernel_size
            state = torch.zeros((B, ks, C), dtype=BF16, device=h1.device)  # F.pad(fused_qkv, (k - s, 0)): zeros on the left if s < k
            n = min(ks, S)
            for b in range(B):
                K.copy2d(proj[b * S + S - n : (b + 1) * S, :C], state[b, ks - n :])
            cache.set_conv_state(i, state)
            y = Q.causal_conv_silu_fwd(proj[:, :C], att.conv1d.weight, B, S)
        qn, kn = Q.l2norm_fwd(y[:, :QK], Hqk, Dk), Q.l2norm_fwd(y[:, QK : 2 * QK], Hqk, Dk)
        beta, alpha = Q.gdn_gates_fwd(proj[:, C + VG : C + VG + Hv], proj[:, C + VG + Hv :], att.log_A, att.dt_bias)
        o, _, fin = Q.gated_delta_rule_fwd(qn, kn, y[:, 2 * QK :], beta, alpha, B, S, Hqk, Hv, Dk, Dv, keep=False, want_state=True,
                                           state=cache.get_recurrent_state(i))
        cache.set_recurrent_state(i, fin)
        mix, _ = Q.gated_rmsnorm_fwd(o, att.post_norm.weight, proj[:, C : C + VG], Hv, Dv, eps=att.post_norm.eps)
        return mix
    H, G, D, QG, KV = ops_q35._att_dims(att)
    proj = st.lin(h1, arena.fused(att.w_queries_gate.weight, att.w_values.weight))
    q, _ = Q.headnorm_rope_fwd(proj[:, :QG], H, D, 2 * D, Q.zc_weight(att.q_norm.scale), st.cos_t, st.sin_t, st.pos, eps=att.q_norm.eps)
    k, _ = Q.headnorm_rope_fwd(proj[:, QG : QG + KV], G, D, D, Q.zc_weight(att.k_norm.scale), st.cos_t, st.sin_t, st.pos, eps=att.k_norm.eps)
    v = proj[:, QG + KV :]
    had_rows = cache.kv_cache.start_pos > 0
    kc, vc, end = cache.append_kv_rows(k, v, i, B, S)
    if decode:  # every cached key is visible: causal for the newest query, and upstream un-masks padded keys
        ctx = attn_decode(q, kc, vc, end, H, G, D)
    else:
        if had_rows:
            raise NotImplementedError("a filled Qwen3_5Cache takes one new token per sequence (use_precomputed_states, qwen3_5_text_model.py:106)")
        ctx, _ = Q.attn_generic_fwd(q, k, v, B, S, H, G, D, key_mask=st.km)
    return Q.sigmoid_gate_fwd(ctx, proj[:, D:QG], H, D, 2 * D)


@torch.no_grad()
def q35_block_cached(blk, h, st):
    """``Qwen3_5TransformerBlock`` on token-major rows ``h`` [B*S, d] with the hybrid cache (reference qwen3_5_text_model.py:296-325)."""
    from . import kernels_q35 as Q
    from . import ops_q35

    arena = ops_q35.arena_for_bf16(blk)
    att, ffn = blk.att, blk.ffn
    h1, _ = K.rmsnorm_fwd(h, Q.zc_weight(blk.norm1.scale), eps=blk.norm1.eps, want_rstd=False)
    mix = q35_mixer_cached(att, arena, h1, st, att.layer_idx)
    h = st.lin(mix, att.out_proj.weight, residual=h)
    h2, _ = K.rmsnorm_fwd(h, Q.zc_weight(blk.norm2.scale), eps=blk.norm2.eps, want_rstd=False)
    gu = st.lin(h2, arena.fused(ffn.lin1.weight, ffn.lin_gate.weight))
    return st.lin(K.swiglu_fwd(gu, ffn.lin1.weight.shape[0]), ffn.lin2.weight, residual=h)


@torch.no_grad()
def qwen35_forward_cached(model, x, cache, attn_mask=None, inputs_embs=None, position_ids=None):
    """Logits (b, s, vocab) of ``Qwen3_5TextModel`` with a ``Qwen3_5Cache`` (reference: qwen3_5_text_model.py:96-191, 206-267, 388-417).

    Prefill (cache empty) runs the sequence kernels of the training path and leaves, per linear layer, the last 4 pre-conv inputs and
    the final recurrent state, per full-attention layer the K / V rows; a one-token step then costs one conv update, ONE recurrence step
    from the carried state (the forward delta-rule kernel with S = 1), resp. one decode-attention pass, with GEMV projections."""
    from . import kernels_q35 as Q

    model._build_arenas()
    if inputs_embs is not None:
        B, S, d = inputs_embs.shape
        h = inputs_embs.reshape(B * S, d).contiguous()
    else:
        L.require_gpu(x)
        B, S = x.shape
        h = K.embedding_fwd(x, model.emb_dict.weight)
    decode = cache.has_previous_state
    if decode and S != 1:
        raise NotImplementedError("with a filled Qwen3_5Cache one new token per sequence is decoded (use_precomputed_states, qwen3_5_text_model.py:106)")
    st = Q35Step(cache, B, S, h.device, model.cos, model.sin, model.mrope_section, attn_mask, position_ids, decode=decode)
    for blk in model.trf_blocks:
        h = q35_block_cached(blk, h, st)
    hn, _ = K.rmsnorm_fwd(h, Q.zc_weight(model.final_norm.scale), eps=model.final_norm.eps, want_rstd=False)
    return st.lin(hn, model.out_head.weight).view(B, S, -1)
