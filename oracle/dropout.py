"""Dropout masks of the HIP path, restated on the CPU.  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The reference's dropout is ``nn.Dropout`` (vit_model.py:146, vit_attention.py:79, vit_transformer_block.py:117,124,
vit_engine.py:51): a Bernoulli(1 - p) keep mask times 1 / (1 - p), drawn from torch's generator.  No two implementations share
torch's stream, so parity for a dropout layer is: (1) the mask is Bernoulli(1 - p) -- checked statistically -- and (2) GIVEN the mask
the arithmetic equals the reference's.  For (2) the product draws its masks from a counter-based generator that this file restates:
Philox4x32-10 (Salmon, Moraes, Dror, Shaw, "Parallel random numbers: as easy as 1, 2, 3", SC'11; the published algorithm -- the same
one cuRAND / rocRAND / torch.cuda use), keyed by a 64-bit seed, counter = (c0, c1, offset_lo, offset_hi):

  element-wise sites (csrc/rope_dropout.hip::dropout_kernel):  element i    -> word i % 4 of Philox(c0 = i // 4 low 32 bits, c1 = high bits)
  attention weights  (csrc/attention_generic.hip::drop4):      (b, h, q, k) -> word k % 4 of Philox(c0 = k // 4, c1 = (b H + h) S + q)

keep iff word >= round(p * 2**32); kept values are multiplied by float32(1 / (1 - p)).
"""

import numpy as np
import torch

_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_LO = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over numpy uint32 arrays c0..c3 (broadcastable); k0, k1 python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint32) for c in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint32(k0 & 0xFFFFFFFF), np.uint32(k1 & 0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _M0 * c0.astype(np.uint64)
            p1 = _M1 * c2.astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _LO).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _LO).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0, k1 = np.uint32(k0 + _W0), np.uint32(k1 + _W1)
    return c0, c1, c2, c3


def threshold(p):
    return min(max(int(float(np.float32(p)) * 4294967296.0 + 0.5), 0), 4294967295)


def inv_keep(p):
    return float(np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def elementwise_multiplier(shape, p, seed, offset):
    """fp32 tensor of ``shape``: 0 or 1 / (1 - p) per element, in the row-major element order the kernel uses."""
    n = int(np.prod(shape))
    g = np.arange((n + 3) // 4, dtype=np.uint64)
    words = philox4x32_10((g & _LO).astype(np.uint32), (g >> np.uint64(32)).astype(np.uint32), np.uint32(offset & 0xFFFFFFFF),
                          np.uint32((offset >> 32) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    bits = np.stack(words, axis=1).reshape(-1)[:n]
    mul = np.where(bits >= np.uint32(threshold(p)), np.float32(inv_keep(p)), np.float32(0.0)).astype(np.float32)
    return torch.from_numpy(mul.reshape(shape))


def attention_multiplier(B, H, S, p, seed, offset):
    """fp32 [B, H, S, S] (query, key): 0 or 1 / (1 - p) on every attention weight."""
    rows = np.arange(B * H * S, dtype=np.uint32).reshape(-1, 1)
    groups = np.arange((S + 3) // 4, dtype=np.uint32).reshape(1, -1)
    words = philox4x32_10(groups, rows, np.uint32(offset & 0xFFFFFFFF), np.uint32((offset >> 32) & 0xFFFFFFFF), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    bits = np.stack(words, axis=2).reshape(B * H * S, -1)[:, :S]
    mul = np.where(bits >= np.uint32(threshold(p)), np.float32(inv_keep(p)), np.float32(0.0)).astype(np.float32)
    return torch.from_numpy(mul.reshape(B, H, S, S))
