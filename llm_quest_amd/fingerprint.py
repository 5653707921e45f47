"""Fingerprint of the kernel sources the headline step runs (csrc/*.hip that the step launches, their shared headers, csrc/Makefile).

Counter files under ``profiles/`` are taken in separate ``rocprofv3 --pmc`` passes and committed; each carries the fingerprint of the
sources it was measured on, written ON THE GPU BOX AT COLLECTION TIME (``tools/collect_evidence.sh`` -> ``stamp.json`` beside the raw
counters, with the sha-256 of the library that was loaded); the aggregators copy that stamp, they never recompute it.  ``bench.py`` prints a
counter-derived figure only when the stamp equals the fingerprint of the sources it runs, so a kernel edit can never leave a stale
``traffic`` in the bench line."""

import hashlib
import json
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
FILES = ("attention.hip", "attention_fwd2.hip", "attn_common.h", "elementwise.hip", "gemm.hip", "norm_rope.hip", "common.h", "Makefile")


def kernel_sources_sha():
    h = hashlib.sha256()
    for n in FILES:
        h.update(n.encode())
        with open(os.path.join(_HERE, "csrc", n), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def library_sha():
    with open(os.path.join(_HERE, "libmi355vlm.so"), "rb") as fh:
        return hashlib.sha256(fh.read()).hexdigest()[:16]


def collection_stamp(evidence_dir):
    """The stamp tools/collect_evidence.sh wrote next to its raw output (kernel_sources_sha, library_sha, git_sha, batch)."""
    with open(os.path.join(evidence_dir, "stamp.json")) as f:
        return json.load(f)


if __name__ == "__main__":  # python -m llm_quest_amd.fingerprint <git sha> <batch>  (collection time, on the GPU box)
    import sys

    print(json.dumps({"kernel_sources_sha": kernel_sources_sha(), "library_sha": library_sha(), "git_sha": sys.argv[1] if len(sys.argv) > 1 else "unknown",
                      "per_gpu_batch": int(sys.argv[2]) if len(sys.argv) > 2 else None}))
