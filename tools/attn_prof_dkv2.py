"""In-kernel cycle stamps of the pipelined dK/dV pass (profiling build -DATTN_ABL=16; MI355_ATTN_ABLATE=16384 selects the kernel).  usage: python tools/attn_prof_dkv2.py [B]"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K

B, S, Hq, Hkv, D = int(sys.argv[1]) if len(sys.argv) > 1 else 160, 709, 16, 8, 128
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
nwg = B * Hkv
n = max(out[5], 1)
print(f"per WG: total {out[0]/nwg:.0f}  iterations {out[5]/nwg:.1f}  block prologue + first A(0) {out[3]/nwg:.0f}  block write-out {out[4]/nwg:.0f}")
print(f"per iteration: {out[2]/n:.0f} cycles = steps 0-24 {out[8]/n:.0f} + wait {out[1]/n:.0f} + barrier {out[6]/n:.0f} + requests {out[7]/n:.0f} + steps 25-63 (not the block's last) {out[9]/n:.0f} + rest")
