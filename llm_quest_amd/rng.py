"""Seed / offset bookkeeping of the counter-based dropout kernels (``mi355_dropout``, ``mi355_attn_dropout_fwd/bwd``).

A dropout site draws ``(seed, offset) = rng.draw()`` in its forward and keeps the pair for its backward, which regenerates the same
Philox mask instead of storing it.  ``seed`` follows ``torch.manual_seed`` (``torch.initial_seed()``); ``offset`` counts the draws
since the seed last changed, so two runs that seed alike and call alike drop alike -- the reproducibility contract of ``nn.Dropout``
under a seeded generator (reference dropout sites: vit_model.py:146, vit_attention.py:79, vit_transformer_block.py:117,124,
vit_engine.py:51).  ``rng.manual(seed, offset)`` pins the next draws explicitly (tests, multi-rank runs that want per-rank streams).
"""

import torch

_state = {"seed": None, "offset": 0}


def manual(seed, offset=0):
    _state["seed"], _state["offset"], _state["pinned"] = int(seed) & 0xFFFFFFFFFFFFFFFF, int(offset), True


def draw():
    """(seed, offset) for one dropout site; advances the offset."""
    if not _state.get("pinned"):
        seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        if seed != _state["seed"]:
            _state["seed"], _state["offset"] = seed, 0
    off = _state["offset"]
    _state["offset"] = off + 1
    return _state["seed"], off


def follow_torch():
    """Back to following ``torch.manual_seed`` after ``manual``."""
    _state["pinned"] = False
    _state["seed"] = None
