"""Seed / offset bookkeeping of the counter-based dropout kernels (``mi355_dropout``, ``mi355_attn_dropout_fwd/bwd``).

A dropout site draws ``(seed, offset) = rng.draw()`` in its forward and keeps the pair for its backward, which regenerates the same
Philox mask instead of storing it (reference dropout sites: vit_model.py:146, vit_attention.py:79, vit_transformer_block.py:117,124,
vit_engine.py:51).

The reference's ``nn.Dropout`` on a device tensor consumes the DEVICE generator and leaves torch's default CPU generator alone, so enabling
dropout must not shift any other consumer of the CPU generator (``DataLoader`` / ``random_split`` shuffles, ``torch.rand`` sampling, later
parameter initialisation).  This module does the same:

  * on a GPU process the pair is read from the current device's default generator -- ``seed`` = its seed, ``offset`` = its Philox offset,
    which is then advanced exactly as a device ``nn.Dropout`` launch would advance it: ``torch.manual_seed(s)`` (same ``s`` too) restarts the
    mask stream, ``torch.cuda.get_rng_state`` / ``set_rng_state`` carry it through checkpoints, and the CPU generator is never touched;
  * without a device (CPU unit tests of the bookkeeping) the offsets come from a DEDICATED ``torch.Generator`` of this module, seeded from
    ``torch.initial_seed()`` and re-seeded when that changes; ``get_state()`` / ``set_state()`` carry it;
  * under data parallelism (``WORLD_SIZE`` > 1) the rank is folded into the seed handed to the kernels, so ranks that were seeded alike still
    draw different masks for their different shards;
  * ``rng.manual(seed, offset)`` pins the next draws explicitly to consecutive offsets (tests); ``follow_torch()`` ends that.
"""

import os

import torch

_state = {"seed": None, "offset": 0, "pinned": False}
_gen = torch.Generator()  # CPU generator of this module only: drawing from it never touches torch's default generator
_gen_seed = None  # the torch.initial_seed() the module generator was last seeded for
_word = None
_M64 = 0xFFFFFFFFFFFFFFFF


def _fold_rank(seed):
    try:
        world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    except ValueError:
        return seed
    return seed if world <= 1 else (seed ^ (0x9E3779B97F4A7C15 * (rank + 1))) & _M64


def _device_generator():
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        return torch.cuda.default_generators[torch.cuda.current_device()]
    return None


def _sync_seed():
    global _gen_seed
    base = torch.initial_seed() & _M64
    if base != _gen_seed:
        _gen_seed = base
        _gen.manual_seed((base * 0x9E3779B97F4A7C15 + 0xD1B54A32D192ED03) & 0x7FFFFFFFFFFFFFFF)
    return base


def manual(seed, offset=0):
    _state["seed"], _state["offset"], _state["pinned"] = int(seed) & _M64, int(offset), True


def draw():
    """(seed, offset) for one dropout site; advances the device generator's offset (the module generator without a device, the pinned
    offset under ``manual``).  No device synchronisation: generator state lives on the host."""
    global _word
    if _state["pinned"]:
        off = _state["offset"]
        _state["offset"] = off + 1
        return _state["seed"], off
    g = _device_generator()
    if g is not None:
        off = int(g.get_offset())
        g.set_offset(off + 4)  # Philox offsets move in units of 4 (one 128-bit block), as the device's own dropout kernels move them
        return _fold_rank(int(g.initial_seed()) & _M64), off // 4
    base = _sync_seed()
    if _word is None:
        _word = torch.empty((), dtype=torch.int64)
    return _fold_rank(base), int(_word.random_(generator=_gen))


def follow_torch():
    """Back to following the generators after ``manual``."""
    _state["pinned"] = False
    _state["seed"] = None


def get_state():
    """Checkpointable state of the host-side mask stream (a GPU process resumes through ``torch.cuda.get_rng_state`` instead)."""
    _sync_seed()
    return {"seed_for": _gen_seed, "generator": _gen.get_state(), "pinned": dict(_state)}


def set_state(st):
    global _gen_seed
    _gen_seed = st["seed_for"]
    _gen.set_state(st["generator"])
    _state.update(st["pinned"])
