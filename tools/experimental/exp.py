"""ctypes binding of the parked kernel experiments (tools/experimental/libmi355exp.so: `make -C tools/experimental`).  NOT part of the product:
libmi355vlm.so does not contain these kernels and include/mi355_vlm.h does not declare them; the tools in this directory time and check them."""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))))
import torch

from llm_quest_amd import _lib as L

_HERE = os.path.dirname(os.path.abspath(__file__))
_c = ctypes
_P, _I, _L, _F = _c.c_void_p, _c.c_int, _c.c_int64, _c.c_float
_lib = None


def load():
    global _lib
    if _lib is None:
        L.load()
        lib = ctypes.CDLL(os.path.join(_HERE, "libmi355exp.so"))
        lib.mi355_exp_attn_fwd2.argtypes = [_I, _I, _I, _I, _I, _P, _L, _P, _L, _P, _L, _P, _L, _P, _P, _I, _F, _P]
        _lib = lib
    return _lib


def _check(rc, name):
    if rc != 0:
        raise RuntimeError(f"{name} failed (rc={rc}): {L.load().mi355_last_error().decode()}")


def attn_fwd2(q, k, v, B, S, Hq, Hkv, D, key_mask=None, causal=True):
    """The one-wave-per-SIMD forward (arguments of kernels.attn_fwd); raises when the shape is not this kernel's."""
    o = torch.empty((B * S, Hq * D), dtype=torch.bfloat16, device=q.device)
    lse = torch.empty((B, Hq, S), dtype=torch.float32, device=q.device)
    rc = load().mi355_exp_attn_fwd2(B, S, Hq, Hkv, D, L.ptr(q), q.stride(0), L.ptr(k), k.stride(0), L.ptr(v), v.stride(0), L.ptr(o), o.stride(0), L.ptr(lse),
                                    L.ptr(key_mask), int(causal), D ** -0.5, L.stream(q.device))
    _check(rc, "mi355_exp_attn_fwd2")
    return o, lse


