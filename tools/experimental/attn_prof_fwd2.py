"""In-kernel cycle stamps of the second-generation attention forward (profiling build -DATTN_ABL=16 of attention_fwd2.hip only):
    make -C llm_quest_amd/csrc FLAGS_attention_fwd2="-fno-slp-vectorize -DATTN_ABL=16" && python tools/attn_prof_fwd2.py
Shares, not lengths: the stamps forbid overlaps the real kernel has."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llm_quest_amd import kernels as K, _lib as L
B, S, Hq, Hkv, D = 64, 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
q, k, v = r(B * S, Hq * D), r(B * S, Hkv * D), r(B * S, Hkv * D)
lib = L.load()
out = (ctypes.c_ulonglong * 32)()
for rep in range(2):
    lib.mi355_debug_prof2(out, 1)
    o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
    lib.mi355_debug_prof2(out, 1)
nwg = 16  # one workgroup in 16 of 256 reports (wave 0)
names = {0: "total", 1: "vmcnt wait", 2: "barrier", 3: "issue DMA", 4: "tile setup", 5: "first-tile S + start", 6: "body (plain)", 14: "body (masks)", 13: "body (last)",
         7: "block prologue", 20: "  of it: Q rows scaled + written", 21: "  first tile: masks + 16 products", 8: "block epilogue", 16: "  of it: next Q request", 17: "  settle",
         18: "  group 0 out", 19: "  group 1 out", 12: "item setup"}
tot = out[0] / nwg
print(f"per workgroup: {tot:.0f} cycles; steps {out[9]/nwg:.1f}, plain/mask bodies {out[10]/nwg:.1f}, last bodies {out[15]/nwg:.1f}, blocks {out[11]/nwg:.1f}")
for i, n in names.items():
    if i:
        print(f"  {n:22s} {out[i]/nwg:9.0f} cycles  {100*out[i]/max(out[0],1):5.1f} %")
nb = max(out[10], 1)
print(f"per plain/mask body: {(out[6]+out[14])/nb:.0f} cycles (2048 of MFMA); per last body {out[13]/max(out[15],1):.0f} (1536)")
