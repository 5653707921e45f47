"""Seed / offset bookkeeping of the counter-based dropout kernels (``mi355_dropout``, ``mi355_attn_dropout_fwd/bwd``).

A dropout site draws ``(seed, offset) = rng.draw()`` in its forward and keeps the pair for its backward, which regenerates the same
Philox mask instead of storing it.  Both numbers come from torch's default generator: ``seed`` is ``torch.initial_seed()`` and ``offset``
is one 63-bit draw from that generator (``Tensor.random_`` on the host), so the stream of masks follows ``torch.manual_seed`` -- seeding
again with the SAME seed restarts it, as it restarts ``nn.Dropout`` -- and ``torch.get_rng_state`` / ``set_rng_state``: a run resumed from a
checkpoint that restores the generator state draws the masks the uninterrupted run would have drawn (reference dropout sites:
vit_model.py:146, vit_attention.py:79, vit_transformer_block.py:117,124, vit_engine.py:51).  ``rng.manual(seed, offset)`` pins the next
draws explicitly to consecutive offsets (tests, multi-rank runs that want per-rank streams); ``follow_torch()`` ends that.
"""

import torch

_state = {"seed": None, "offset": 0, "pinned": False}
_word = None


def manual(seed, offset=0):
    _state["seed"], _state["offset"], _state["pinned"] = int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset), True


def draw():
    """(seed, offset) for one dropout site; advances the generator (or the pinned offset)."""
    global _word
    if _state["pinned"]:
        off = _state["offset"]
        _state["offset"] = off + 1
        return _state["seed"], off
    if _word is None:
        _word = torch.empty((), dtype=torch.int64)
    return torch.initial_seed() & 0xFFFFFFFFFFFFFFFF, int(_word.random_())  # host generator: no device sync


def follow_torch():
    """Back to following torch's default generator after ``manual``."""
    _state["pinned"] = False
    _state["seed"] = None
