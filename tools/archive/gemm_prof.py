"""In-kernel cycle stamps of the alternating-group GEMM loop (profiling build -DGEMM_PROF=1 only)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K, _lib as L
lib = L.load()
out = (ctypes.c_ulonglong * 32)()
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
w = r(4096, 4096)
for _ in range(100): w @ w
names = ["frag reads issue", "DMA issue", "vmcnt wait", "lgkmcnt wait", "barrier 1", "MFMA cluster", "barrier 2"]
for name, form, sa, sb in (("NT 45376x6144x1024", L.GEMM_NT, (45376, 1024), (6144, 1024)), ("NN 45376x1024x6144", L.GEMM_NN, (45376, 6144), (6144, 1024)),
                           ("NT 4096x4096x8192", L.GEMM_NT, (4096, 8192), (4096, 8192)), ("TN 4096x4096x8192", L.GEMM_TN, (8192, 4096), (8192, 4096))):
    a, b = r(*sa), r(*sb)
    o = K.gemm(form, a, b, tile=3, allow_split_k=False)
    lib.mi355_debug_gemm_prof(out, 1)
    K.gemm(form, a, b, out=o, tile=3, allow_split_k=False)
    lib.mi355_debug_gemm_prof(out, 1)
    for g, off in (("early", 0), ("late", 16)):
        ph = max(out[off + 8], 1)
        print(f"{name:20s} {g:5s} per phase: " + "  ".join(f"{n} {out[off + i] / ph:6.0f}" for i, n in enumerate(names)) + f"  | loop total {out[off + 7] / ph:6.0f}")
