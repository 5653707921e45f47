"""CPU oracle for the VLM forward/backward hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is part of the product:
only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import it, and only as the checker / reported CPU baseline.
The product path (``llm_quest_amd``) never falls back to this code.

What it is: a functional (state-dict in, tensors out) PyTorch-CPU restatement of
the reference algorithms on the path named by BASELINE.json's north_star.  Each
function cites the reference file:line it follows (paths relative to the
reference checkout).  All arithmetic runs through PyTorch ATen on the CPU
(reference pin: ``torch>=2.8.0``, pyproject.toml:16); the fixtures were
produced with torch 2.10.0+rocm7.0 (CPU build) in the build container.

Parity pinning: the reference ships no tests/golden vectors for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference
itself *run in the build container*: ``oracle/gen_golden.py`` imports the
reference from /root/reference, runs seeded tiny models / per-op cases, and
writes inputs + expected outputs + gradients to ``tests/golden/*.safetensors``.
``tests/test_oracle_golden.py`` checks every oracle function against those.
"""

from . import dropout, index_ops, ops, models, qwen3_5, qwen3_5_text  # noqa: F401
