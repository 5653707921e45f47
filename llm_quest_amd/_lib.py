"""ctypes binding of libmi355vlm.so (C ABI in include/mi355_vlm.h).

The library is the product: if it is missing the package raises at first use -- there is no CPU or eager-PyTorch
fallback.  ``build()`` compiles it in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""

import ctypes
import os
import subprocess

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MI355_LIB_PATH") or os.path.join(_HERE, "libmi355vlm.so")  # override: profiling builds only
CSRC = os.path.join(_HERE, "csrc")

DT_BF16, DT_F32, DT_SPLIT3 = 0, 1, 2  # DT_SPLIT3: an fp32 value as bf16 [hi | lo | hi] column blocks (include/mi355_vlm.h)
GEMM_NT, GEMM_NN, GEMM_TN = 0, 1, 2
EPI_NONE, EPI_GELU, EPI_SWIGLU_BWD, EPI_SWIGLU_FWD = 0, 1, 2, 3
EPI_GELU_DUAL_ERF, EPI_GELU_DUAL_TANH, EPI_GELU_BWD_ERF, EPI_GELU_BWD_TANH = 4, 5, 6, 7
ATTN_DELTA_READY = 0x1000000  # MI355_ATTN_DELTA_READY

_c = ctypes
_P, _I, _L, _F, _U = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float, _c.c_uint64

# name -> argtypes; every entry point returns int.  Must list exactly what include/mi355_vlm.h declares.
SIGNATURES = {
    "mi355_gemm_bf16": [_I, _L, _L, _L, _P, _L, _P, _L, _P, _L, _I, _P, _P, _L, _I, _P, _L, _I, _P],
    "mi355_gemm_bf16_grouped": [_I, _I, _P, _I, _I, _P],
    "mi355_gemm_bf16_attn_delta": [_L, _L, _L, _P, _L, _P, _L, _P, _L, _P, _L, _I, _I, _I, _P, _P, _P, _P, _P],
    "mi355_adamw": [_L, _P, _I, _P, _I, _P, _P, _F, _F, _F, _F, _F, _I, _P, _F, _P],
    "mi355_colsum": [_L, _L, _P, _I, _L, _P, _I, _P],
    "mi355_rmsnorm_fwd": [_L, _I, _P, _P, _P, _P, _F, _P],
    "mi355_rmsnorm_bwd": [_L, _I, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "mi355_reduce_rows_f32": [_I, _L, _P, _P, _I, _I, _P],
    "mi355_qknorm_rope_fwd": [_L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P],
    "mi355_qknorm_rope_bwd": [_L, _I, _I, _I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P],
    "mi355_swiglu_fwd": [_L, _I, _P, _P, _P],
    "mi355_swiglu_bwd": [_L, _I, _P, _P, _P, _P],
    "mi355_attn_fwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _F, _P],
    "mi355_attn_bwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _P, _I, _F, _P],
    "mi355_attn_bwd_ws": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _P, _I, _F, _P, _L, _P],
    "mi355_attn_bwd_qnorm": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _I, _F, _P, _L,
                             _P, _L, _P, _P, _P, _P, _P, _I, _P, _L, _P, _P, _P],
    "mi355_cross_entropy": [_L, _L, _P, _L, _P, _P, _P, _P, _P],
    "mi355_ce_finalize": [_L, _P, _P, _P, _P],
    "mi355_embedding_fwd": [_L, _I, _L, _P, _P, _P, _L, _P],
    "mi355_embedding_bwd": [_L, _I, _L, _P, _P, _L, _P, _P],
    "mi355_embedding_bwd_sorted": [_L, _I, _L, _P, _P, _P, _L, _F, _P, _L, _I, _P, _L, _P],
    "mi355_copy2d": [_L, _L, _P, _L, _P, _L, _P],
    "mi355_transpose_bf16": [_L, _L, _P, _L, _P, _L, _P],
    "mi355_patchify": [_I, _I, _I, _I, _I, _P, _P, _I, _P],
    "mi355_layernorm_fwd": [_L, _I, _P, _P, _P, _P, _I, _P, _P, _F, _I, _P],
    "mi355_layernorm_bwd": [_L, _I, _P, _P, _P, _P, _P, _I, _P, _P, _P, _I, _F, _I, _P],
    "mi355_cast": [_L, _P, _I, _P, _I, _P],
    "mi355_vit_embed_assemble": [_I, _I, _I, _P, _P, _P, _P, _P],
    "mi355_split3_bf16": [_L, _I, _P, _L, _P, _I, _P],
    "mi355_attn_f32_fwd": [_I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _F, _P],
    "mi355_sumsq": [_L, _P, _I, _P, _P, _P],
    "mi355_clip_scale": [_L, _P, _I, _P, _F, _P],
    "mi355_add_f32_to_bf16": [_L, _P, _P, _P, _P],
    "mi355_scale_bf16": [_L, _P, _P, _P, _P],
    "mi355_gelu_fwd": [_L, _P, _P, _I, _P],
    "mi355_gelu_bwd": [_L, _P, _P, _P, _I, _P],
    "mi355_patchify3d": [_I, _I, _I, _I, _I, _I, _I, _P, _P, _I, _P],
    "mi355_merge_patches": [_L, _I, _I, _I, _L, _P, _P, _I, _P],
    "mi355_scatter_rows": [_L, _L, _P, _P, _P, _P, _P, _P, _I, _P],
    # Qwen3.5 text stack (csrc/qwen35.hip, csrc/attention_generic.hip)
    "mi355_zc_weight": [_L, _P, _P, _P],
    "mi355_mrope_table": [_L, _I, _L, _P, _P, _P, _I, _I, _P, _P, _P],
    "mi355_rowmask": [_L, _I, _P, _P, _P, _P],
    "mi355_headnorm_rope_fwd": [_L, _I, _I, _I, _P, _L, _L, _P, _P, _P, _P, _P, _P, _F, _P],
    "mi355_headnorm_rope_bwd": [_L, _I, _I, _I, _P, _L, _L, _P, _P, _P, _P, _P, _P, _P, _L, _L, _P, _I, _P],
    "mi355_sigmoid_gate_fwd": [_L, _I, _I, _P, _P, _L, _L, _P, _P],
    "mi355_sigmoid_gate_bwd": [_L, _I, _I, _P, _P, _L, _L, _P, _P, _P, _L, _L, _P],
    "mi355_attn_generic_fwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _F, _P],
    "mi355_attn_generic_bwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _P, _F, _P],
    "mi355_gdn_gates_fwd": [_L, _I, _P, _P, _L, _P, _P, _P, _P, _P],
    "mi355_gdn_gates_bwd": [_L, _I, _P, _P, _L, _P, _P, _P, _P, _P, _P, _L, _P, _I, _P],
    "mi355_causal_conv_silu_fwd": [_I, _I, _I, _I, _P, _L, _P, _P, _P],
    "mi355_causal_conv_silu_bwd": [_I, _I, _I, _I, _P, _L, _P, _P, _P, _P, _L, _P, _I, _P],
    "mi355_l2norm_fwd": [_L, _I, _I, _P, _L, _P, _P],
    "mi355_l2norm_bwd": [_L, _I, _I, _P, _L, _P, _P, _L, _P],
    "mi355_gated_delta_rule_fwd": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P],
    "mi355_causal_conv_silu_step": [_I, _I, _I, _P, _L, _P, _P, _P, _P],
    "mi355_gated_delta_rule_bwd": [_I, _I, _I, _I, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _P, _P, _P, _L, _P, _P, _P],
    "mi355_gated_rmsnorm_fwd": [_L, _I, _I, _P, _P, _P, _L, _P, _P, _F, _P],
    "mi355_gated_rmsnorm_bwd": [_L, _I, _I, _P, _P, _P, _L, _P, _P, _P, _P, _L, _P, _I, _P],
    # input pipeline (csrc/pipeline.hip)
    "mi355_resize_h_u8": [_I, _I, _I, _I, _P, _L, _P, _P, _I, _P, _P],
    "mi355_resize_v_normalize": [_I, _I, _I, _I, _P, _P, _P, _I, _P, _P, _P, _P],
    "mi355_pad_tokens": [_I, _I, _P, _P, _L, _P, _P, _P],
    # KV-cache decoding (csrc/decode.hip)
    "mi355_gemv_bf16": [_I, _L, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P],
    "mi355_gemv_bf16_pro": [_I, _L, _I, _P, _L, _I, _P, _F, _P, _L, _P, _L, _P, _L, _P],
    "mi355_attn_decode_qkv": [_I, _I, _I, _I, _P, _L, _P, _P, _P, _P, _P, _P, _P, _L, _L, _I, _P, _P, _P, _L, _P, _F, _F, _I, _P, _P],
    "mi355_decode_advance": [_I, _P, _P, _P, _P, _P, _P],
    "mi355_attn_decode": [_I, _I, _I, _I, _P, _P, _P, _L, _L, _I, _P, _P, _L, _P, _F, _I, _P, _P],
    "mi355_kv_append": [_I, _I, _P, _L, _P, _L, _P, _P, _L, _L, _I, _P, _P],
    "mi355_argmax_rows": [_L, _L, _P, _L, _P, _P, _P],
    # measurement aid (csrc/probe.hip)
    "mi355_mfma_pipe_probe": [_I, _I, _P, _P],
    # stand-alone RoPE and dropout (csrc/rope_dropout.hip, csrc/attention_generic.hip)
    "mi355_rope_apply": [_I, _I, _I, _I, _I, _P, _I, _L, _L, _L, _P, _P, _L, _P, _P, _L, _L, _L, _I, _P],
    "mi355_dropout": [_L, _P, _I, _P, _P, _I, _F, _U, _U, _P],
    "mi355_attn_dropout_fwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _I, _F, _F, _U, _U, _P],
    "mi355_attn_generic_dropout_fwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _F, _F, _U, _U, _P],
    "mi355_attn_generic_dropout_bwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _P, _F, _F, _U, _U, _P],
    "mi355_attn_dropout_bwd": [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _P, _L, _P, _L, _P, _L, _I, _F, _F, _U, _U, _P],
}
# size / constant queries: no stream argument, the return value is the answer (name -> (argtypes, restype))
QUERIES = {
    "mi355_gated_delta_rule_chunk": ([], _I),
    "mi355_gated_delta_rule_bwd_workspace_bytes": ([_I, _I, _I, _I, _I], _L),
    "mi355_attn_bwd_workspace_bytes": ([_I, _I, _I, _I], _L),
    "mi355_attn_bwd_workspace_rowconst_offset": ([_I, _I, _I, _I, _I], _L),
    "mi355_attn_bwd_qnorm_partials": ([_I, _I, _I], _L),
    "mi355_embedding_bwd_sorted_workspace_bytes": ([_L, _I], _L),
}

_lib = None


def build(verbose=False):
    """Compile libmi355vlm.so in-tree (hipcc --offload-arch=gfx950)."""
    cmd = ["make", "-C", CSRC, "-j8"]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError("building libmi355vlm.so failed:\n" + res.stdout[-4000:] + res.stderr[-4000:])
    if verbose:
        print(res.stdout[-2000:])
    return LIB_PATH


def load():
    """Load the shared library (no GPU needed) and bind every declared symbol."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: the HIP extension is the only implementation of this path "
            "(no CPU fallback). Build it with `python -c 'import __graft_entry__ as g; g.build()'`."
        )
    lib = ctypes.CDLL(LIB_PATH)
    lib.mi355_last_error.restype = ctypes.c_char_p
    lib.mi355_last_error.argtypes = []
    lib.mi355_abi_version.restype = ctypes.c_int
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.argtypes = argtypes
        fn.restype = ctypes.c_int
    for name, (argtypes, restype) in QUERIES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    _lib = lib
    return lib


class GemmProblem(_c.Structure):
    """struct mi355_gemm_problem (include/mi355_vlm.h)."""

    _fields_ = [("M", _L), ("N", _L), ("K", _L), ("A", _P), ("lda", _L), ("B", _P), ("ldb", _L), ("C", _P), ("ldc", _L),
                ("residual", _P), ("ldr", _L)]


def ptr(t):
    return None if t is None else t.data_ptr()


import threading

_tls = threading.local()  # .device: device of the tensors checked by this THREAD's last require_gpu() -- the launches that follow run there
# (autograd runs one thread per device: a module-level global would let one thread's require_gpu steer another thread's launch)


def stream(device=None):
    return torch.cuda.current_stream(device).cuda_stream


def call(name, *args):
    """Invoke an entry point on the current HIP stream OF THE DEVICE THE OPERANDS LIVE ON (``require_gpu`` recorded it); raise
    RuntimeError with the library's message on failure.  Without the guard a model on cuda:1 under a current device of cuda:0
    would be enqueued on cuda:0's stream with cuda:1's pointers.  The recorded device stays in force until the thread's NEXT
    ``require_gpu``: a wrapper that launches several kernels (a backward followed by its partial-sum reductions) checks its operands
    once and every launch of it lands on their device."""
    lib = load()
    dev = getattr(_tls, "device", None)
    if dev is not None and dev.index is not None and dev.index != torch.cuda.current_device():
        with torch.cuda.device(dev):
            rc = getattr(lib, name)(*args, stream(dev))
    else:
        rc = getattr(lib, name)(*args, stream(dev))
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {lib.mi355_last_error().decode()}")


def require_gpu(*tensors):
    """Every operand must be a HIP tensor, and all of them on ONE device; remembers that device for the ``call``s that follow (until the
    thread's next ``require_gpu``; a call with no tensor operand at all clears it: the current device's stream is used)."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError(
                "llm_quest_amd ops run only on an MI355X (HIP) device; got a CPU tensor. "
                "There is no CPU fallback for this path."
            )
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise RuntimeError(f"llm_quest_amd ops: operands live on different devices ({dev} and {t.device})")
    _tls.device = dev


def dt_code(dtype):
    if dtype == torch.bfloat16:
        return DT_BF16
    if dtype == torch.float32:
        return DT_F32
    raise TypeError(f"unsupported dtype {dtype}")
