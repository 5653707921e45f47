"""In-kernel cycle stamps of the attention forward (profiling build -DATTN_ABL=16 only; MI355_ATTN_ABLATE=4096 turns the forward's stamps on)."""
import sys, os, ctypes
os.environ["MI355_ATTN_ABLATE"] = "4096"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K, _lib as L
B, S, Hq, Hkv, D = 64, 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
q, k, v = r(B * S, Hq * D), r(B * S, Hkv * D), r(B * S, Hkv * D)
lib = L.load()
out = (ctypes.c_ulonglong * 16)()
for rep in range(2):
    lib.mi355_debug_prof(out, 1)
    o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
    lib.mi355_debug_prof(out, 1)
nwg = B * Hq * ((S + 127) // 128) // 64  # one workgroup in 64 reports
print(f"per WG (wave 0): total {out[0]/nwg:.0f} cycles, tiles {out[7]/nwg:.1f} (active {out[6]/nwg:.1f})")
t, a = max(out[7], 1), max(out[6], 1)
print(f"per tile: sync {out[1]/t:.0f}  issue {out[2]/t:.0f}   per ACTIVE tile: S phase {out[3]/a:.0f}  softmax {out[4]/a:.0f}  PV phase {out[5]/a:.0f}")
