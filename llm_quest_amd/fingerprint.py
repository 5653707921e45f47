"""Fingerprint of the kernel sources the headline step runs: the csrc/*.hip files the step launches, their shared headers, and the compile
flags those files are built with (the ``CXXFLAGS`` / ``FLAGS_<file>`` lines of csrc/Makefile -- NOT the Makefile as a whole: adding an
unrelated source to ``SRCS`` must not invalidate the counters of kernels that did not change, which is what cost round 3 its ``traffic``).

Counter files under ``profiles/`` are taken in separate ``rocprofv3 --pmc`` passes and committed; each carries the fingerprint of the
sources it was measured on, written ON THE GPU BOX AT COLLECTION TIME (``tools/collect_evidence.sh`` -> ``stamp.json`` beside the raw
counters, with the sha-256 of the library that was loaded); the aggregators copy that stamp, they never recompute it.  ``bench.py`` prints a
counter-derived figure only when the stamp equals the fingerprint of the sources it runs, so a kernel edit can never leave a stale
``traffic`` in the bench line; ``tests/test_zz_evidence_cpu.py`` fails while the committed stamp of ``EVIDENCE_ROUND`` is stale at HEAD."""

import hashlib
import json
import os
import re

_HERE = os.path.dirname(os.path.abspath(__file__))
EVIDENCE_ROUND = "r06"  # profiles/<EVIDENCE_ROUND>_* are the counter files bench.py reads
FILES = ("attention.hip", "attn_common.h", "elementwise.hip", "gemm.hip", "norm_rope.hip", "common.h")
_FLAG_LINES = re.compile(r"^(CXXFLAGS|ARCH|FLAGS_(?:attention|elementwise|gemm|norm_rope))\s*\??=")


def compile_flags():
    """The Makefile lines that decide how FILES are compiled, in file order (variable name included, whitespace normalised)."""
    out = []
    with open(os.path.join(_HERE, "csrc", "Makefile")) as fh:
        for line in fh:
            if _FLAG_LINES.match(line):
                out.append(" ".join(line.split()))
    return "\n".join(out)


def kernel_sources_sha():
    h = hashlib.sha256()
    for n in FILES:
        h.update(n.encode())
        with open(os.path.join(_HERE, "csrc", n), "rb") as fh:
            h.update(fh.read())
    h.update(compile_flags().encode())
    return h.hexdigest()[:16]


def all_sources_sha():
    """Every kernel source of the library (all csrc/*.hip and *.h) + every flag line: the stamp the counter files of the OTHER configurations carry
    (``bench.py --config 2 / 3 / 5`` run kernels outside FILES: qwen35.hip, attention_generic.hip, rope_dropout.hip, ...)."""
    h = hashlib.sha256()
    d = os.path.join(_HERE, "csrc")
    for n in sorted(f for f in os.listdir(d) if f.endswith((".hip", ".h")) and not f.startswith("_")):
        h.update(n.encode())
        with open(os.path.join(d, n), "rb") as fh:
            h.update(fh.read())
    with open(os.path.join(d, "Makefile")) as fh:
        h.update("\n".join(" ".join(line.split()) for line in fh if re.match(r"^(CXXFLAGS|ARCH|FLAGS_\w+)\s*\??=", line)).encode())
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

    print(json.dumps({"kernel_sources_sha": kernel_sources_sha(), "all_sources_sha": all_sources_sha(), "library_sha": library_sha(), "git_sha": sys.argv[1] if len(sys.argv) > 1 else "unknown",
                      "per_gpu_batch": int(sys.argv[2]) if len(sys.argv) > 2 else None}))
