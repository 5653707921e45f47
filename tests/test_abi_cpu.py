"""CPU-side checks of the C-ABI boundary: the library loads without a GPU and exports exactly what
include/mi355_vlm.h declares; the product path refuses CPU tensors instead of falling back."""

import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "mi355_vlm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\*)\s+(mi355_\w+)\s*\(", text)))


def test_header_and_binding_agree():
    from llm_quest_amd import _lib

    declared = set(header_functions())
    bound = set(_lib.SIGNATURES) | set(_lib.QUERIES) | {"mi355_last_error", "mi355_abi_version"}
    assert declared == bound, f"header-only: {declared - bound}; binding-only: {bound - declared}"


def test_library_loads_and_exports_every_symbol():
    from llm_quest_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.mi355_abi_version() == 1


def test_argument_counts_match_header():
    from llm_quest_amd import _lib

    text = open(os.path.join(ROOT, "include", "mi355_vlm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    queries = {n: a for n, (a, _) in _lib.QUERIES.items()}
    for name, argtypes in {**_lib.SIGNATURES, **queries}.items():
        m = re.search(r"\b" + name + r"\s*\((.*?)\)\s*;", text, flags=re.S)
        assert m, name
        nargs = len([a for a in m.group(1).split(",") if a.strip() and a.strip() != "void"])
        assert nargs == len(argtypes) + 0, f"{name}: header has {nargs} args, binding {len(argtypes)}"


def test_host_side_validation_rejects_bad_calls_without_touching_the_gpu():
    """Errors are returned (not faults): the ABI validates before launching."""
    from llm_quest_amd import _lib

    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    lib = _lib.load()
    rc = lib.mi355_gemm_bf16(7, 1, 1, 8, None, 8, None, 8, None, 8, 0, None, None, 0, 0, None, 0, 0, None)
    assert rc != 0 and b"bad form" in lib.mi355_last_error()
    rc = lib.mi355_attn_fwd(1, 4, 2, 1, 96, None, 0, None, 0, None, 0, None, 0, None, None, 1, 1.0, None)
    assert rc != 0 and b"head_dim" in lib.mi355_last_error()


def test_host_side_of_every_entry_point_is_clean_under_address_sanitizer():
    """SURVEY.md section 5 (sanitizers, CPU build only): the AddressSanitizer build of the library (`make asan`: host code instrumented,
    device code untouched) takes degenerate calls of EVERY entry point -- null pointers, zero / negative / huge sizes, odd enums -- in a
    child process with the ASAN runtime preloaded (tools/asan_abi_probe.py).  Each call must return a code; ASAN must report nothing."""
    import glob
    import shutil
    import subprocess
    import sys

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    rts = glob.glob("/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so")
    if not os.path.exists(hipcc) or not rts:
        pytest.skip("hipcc or its ASAN runtime not available")
    csrc = os.path.join(ROOT, "llm_quest_amd", "csrc")
    res = subprocess.run(["make", "-C", csrc, "-j8", "asan"], capture_output=True, text=True, timeout=1500)  # incremental: seconds when up to date
    assert res.returncode == 0, res.stderr[-2000:]
    lib = os.path.join(ROOT, "llm_quest_amd", "libmi355vlm_asan.so")
    env = dict(os.environ, LD_PRELOAD=rts[0], ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=23")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "asan_abi_probe.py"), lib], capture_output=True, text=True, env=env, timeout=600)
    assert res.returncode == 0 and "no fault" in res.stdout, (res.stdout[-500:], res.stderr[-3000:])
    assert "AddressSanitizer" not in res.stderr, res.stderr[-3000:]


def test_no_cpu_fallback():
    from llm_quest_amd import kernels

    x = torch.zeros(4, 1024, dtype=torch.bfloat16)
    w = torch.ones(1024, dtype=torch.bfloat16)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        kernels.rmsnorm_fwd(x, w)


def test_attention_backward_owned_agprs_are_untouched_by_the_compiler(tmp_path):
    """attention.hip owns the top AGPRs of its backward kernels by name (gradient tiles + resident operands).  The generated
    code must not name a register of those ranges outside the kernels' own asm statements (tools/audit_agpr.py)."""
    import importlib.util
    import shutil
    import subprocess

    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, "llm_quest_amd", "csrc", "attention.hip")
    out = tmp_path / "attention.s"
    res = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", "-S", "--cuda-device-only",
                          "-o", str(out), src], capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-2000:]
    spec = importlib.util.spec_from_file_location("audit_agpr", os.path.join(ROOT, "tools", "audit_agpr.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, seen = mod.audit(str(out))
    assert not bad, bad[:5]
    assert set(seen) == set(mod.OWNED), seen


def test_integration_md_stubs_match_the_binding_table():
    """Every `_lib.<name>.argtypes = [...]` line of INTEGRATION.md (the stubs a reference maintainer would copy) must list exactly the
    argument types of `_lib.SIGNATURES` -- the doc cannot drift from the header again (round 2: a 16-argument stub of a 17-argument call)."""
    import ctypes

    from llm_quest_amd import _lib

    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    letters = {"P": ctypes.c_void_p, "I": ctypes.c_int, "L": ctypes.c_int64, "F": ctypes.c_float,
               "ctypes.c_uint64": ctypes.c_uint64, "U": ctypes.c_uint64}
    stubs = re.findall(r"_lib\.(mi355_\w+)\.argtypes\s*=\s*\[(.*?)\]", text)
    assert len(stubs) >= 4, "INTEGRATION.md lost its ctypes stubs"
    for name, body in stubs:
        assert name in _lib.SIGNATURES, f"INTEGRATION.md binds {name}, which the library does not declare"
        doc = [letters[a.strip()] for a in body.split(",") if a.strip()]
        assert doc == list(_lib.SIGNATURES[name]), f"INTEGRATION.md stub of {name}: {len(doc)} args vs {len(_lib.SIGNATURES[name])} in the binding table"
    # call sites in the doc pass as many arguments as the stub declares
    for name, body in stubs:
        m = re.search(r"_lib\." + name + r"\((.*?)\)\)", text, flags=re.S)
        if not m:
            continue
        depth, nargs, cur = 0, 0, ""
        for ch in m.group(1):
            if ch in "([":
                depth += 1
            elif ch in ")]":
                depth -= 1
            if ch == "," and depth == 0:
                nargs += 1
                cur = ""
            else:
                cur += ch
        nargs += 1 if cur.strip() else 0
        assert nargs == len(_lib.SIGNATURES[name]), f"INTEGRATION.md calls {name} with {nargs} arguments, the ABI takes {len(_lib.SIGNATURES[name])}"
