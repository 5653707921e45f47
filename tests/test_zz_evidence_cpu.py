"""The committed counter evidence must belong to the kernels at HEAD (runs last in the CPU suite: the file name sorts behind every other test).

``bench.py`` prints ``roofline.traffic`` only when ``profiles/<round>_pmc_tcc_step.json`` carries the fingerprint of the kernel sources it runs
(llm_quest_amd/fingerprint.py).  Round 3 lost that field in the driver's record because a commit after the last collection changed the
fingerprint; this test turns the same situation into a red suite: after the round's last kernel-affecting edit, re-run
``tools/collect_evidence.sh`` on the GPU box and ``tools/aggregate_evidence.sh`` here, and commit the new profiles."""

import json
import os

from llm_quest_amd import fingerprint as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load(name):
    with open(os.path.join(ROOT, "profiles", f"{F.EVIDENCE_ROUND}_{name}")) as f:
        return json.load(f)


def test_the_fingerprint_ignores_everything_but_the_steps_sources_and_flags(tmp_path):
    flags = F.compile_flags()
    assert "CXXFLAGS" in flags and "FLAGS_attention" in flags and "SRCS" not in flags and "ASAN" not in flags
    assert len(F.kernel_sources_sha()) == 16 and F.kernel_sources_sha() == F.kernel_sources_sha()


def test_committed_counter_files_were_measured_on_the_kernel_sources_at_head():
    head = F.kernel_sources_sha()
    stamp = _load("stamp.json")
    assert stamp["kernel_sources_sha"] == head, (
        f"profiles/{F.EVIDENCE_ROUND}_stamp.json was taken on kernel sources {stamp['kernel_sources_sha']}, HEAD has {head}: re-collect the evidence "
        "(tools/collect_evidence.sh on the GPU box, tools/aggregate_evidence.sh here) as the LAST kernel-affecting act of the round")
    for name in ("pmc_tcc_step.json", "pmc_tcc_gemm.json", "pmc_sq_counters.json"):
        d = _load(name)
        assert d["kernel_sources_sha"] == head, f"profiles/{F.EVIDENCE_ROUND}_{name} is stale ({d['kernel_sources_sha']} != {head})"
        assert d["library_sha"] == stamp["library_sha"]
