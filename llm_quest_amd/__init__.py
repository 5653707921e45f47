"""llm_quest_amd -- MI355X-native (gfx950) implementation of LLM-quest's VLM forward/backward hot path.

Python host code mirrors the reference's module / engine API (same class names, constructor-dict keys, forward
signatures, state_dict keys); every op underneath is a hand-written HIP kernel reached through the C ABI of
``libmi355vlm.so`` (``include/mi355_vlm.h``).  There is no CPU fallback for the hot path.
"""

from . import _lib  # noqa: F401

__all__ = ["_lib"]
