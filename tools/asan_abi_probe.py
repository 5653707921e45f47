"""Drive the host side of every C-ABI entry point with degenerate arguments (null pointers, zero and negative sizes, bad enums).
Run by tests/test_abi_cpu.py against the AddressSanitizer build of the library (make -C llm_quest_amd/csrc asan) with the ASAN
runtime preloaded: every call must come back with a return code (an error, or 0 for an empty problem) -- never a fault, and ASAN
must have nothing to report about the validation code, the parameter blocks or the error-message buffer.  No GPU is needed: a call
that passes validation on this box fails at the launch with a HIP error code, which the ABI returns like any other.

usage: LD_PRELOAD=<libclang_rt.asan> python tools/asan_abi_probe.py <libmi355vlm_asan.so>"""
import ctypes
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def signatures():
    # the binding table without importing the package (which imports torch: too slow and too noisy under ASAN)
    src = open(os.path.join(ROOT, "llm_quest_amd", "_lib.py")).read()
    head = src[: src.index("def ")]
    head = head.replace("import torch\n", "")
    ns = {}
    ns["__file__"] = os.path.join(ROOT, "llm_quest_amd", "_lib.py")
    exec(compile(head, "_lib_head", "exec"), ns)
    return ns["SIGNATURES"], ns["QUERIES"]


def main(path):
    lib = ctypes.CDLL(path)
    lib.mi355_last_error.restype = ctypes.c_char_p
    sigs, queries = signatures()
    patterns = {"zeros": lambda t, i: t(0) if t is not ctypes.c_void_p else None,
                "negative": lambda t, i: t(-1) if t in (ctypes.c_int, ctypes.c_int64) else (t(0) if t is not ctypes.c_void_p else None),
                "huge": lambda t, i: t(2**30 + 7) if t in (ctypes.c_int, ctypes.c_int64) else (t(0) if t is not ctypes.c_void_p else None),
                "odd": lambda t, i: t(3 + i) if t in (ctypes.c_int, ctypes.c_int64) else (t(1) if t is not ctypes.c_void_p else None)}
    calls = errors = 0
    for name, argtypes in sigs.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, ctypes.c_int
        for label, make in patterns.items():
            rc = fn(*[make(t, i) for i, t in enumerate(argtypes)])
            calls += 1
            if rc != 0:
                errors += 1
                msg = lib.mi355_last_error()
                assert msg is not None and len(msg) > 0, f"{name}({label}) returned {rc} with an empty error message"
    for name, (argtypes, restype) in queries.items():
        fn = getattr(lib, name)
        fn.argtypes, fn.restype = argtypes, restype
        for label, make in patterns.items():
            fn(*[make(t, i) for i, t in enumerate(argtypes)])
            calls += 1
    print(f"asan-probe: {calls} calls, {errors} refused, no fault")


if __name__ == "__main__":
    main(sys.argv[1])
