"""In-kernel cycle stamps of the dK/dV pass (profiling build -DATTN_ABL=16 only)."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K, _lib as L
B, S, Hq, Hkv, D = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
qkv = r(B * S, (Hq + 2 * Hkv) * D)
q, k, v = r(B * S, Hq * D), r(B * S, Hkv * D), qkv[:, (Hq + Hkv) * D:]
do = r(B * S, Hq * D)
dq, dk, dqkv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(qkv)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
lib = L.load()
out = (ctypes.c_ulonglong * 16)()
for rep in range(2):
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dqkv[:, (Hq + Hkv) * D:], key_mask=km, causal=True)
    lib.mi355_debug_prof(out, 1)
names = ["wg total", "vmcnt wait", "tile body", "block prologue", "block epilogue", "tiles", "barrier", "issue"]
nwg = B * Hkv
for i, n in enumerate(names): print(f"{n:16s} {out[i]/nwg:12.0f} per WG")
print("segments per tile: A0 %.0f  A1|B0 %.0f  C0|B1 %.0f  C1 %.0f" % tuple(out[8 + i] / max(out[5], 1) for i in range(4)))
print(f"per tile: body {out[2]/max(out[5],1):.0f} cycles, wait {out[1]/max(out[5],1):.0f}, barrier {out[6]/max(out[5],1):.0f}, issue {out[7]/max(out[5],1):.0f}")
