"""Same-process A/B of the lean attention forward: next-tile DMA pieces spread over the S product (default) against one burst behind the barrier (ablation bit 9).
usage: python tools/ab_fwd_spread.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
q, k, v = r(B * S, Hq * D), r(B * S, Hkv * D), r(B * S, Hkv * D)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
def run(bit, n, mask):
    keep = K._ATTN_ABLATE
    K._ATTN_ABLATE = keep | (bit << 8)
    try:
        for _ in range(5): K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=mask, causal=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=mask, causal=True)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    finally:
        K._ATTN_ABLATE = keep
for rnd in range(4):
    print(f"round {rnd}: burst {run(512, 100, km):6.1f} us  spread {run(0, 100, km):6.1f} us   (no mask: burst {run(512, 100, None):6.1f}  spread {run(0, 100, None):6.1f})", flush=True)
o1, l1 = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
K._ATTN_ABLATE |= 512 << 8
o0, l0 = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
print("bit-identical:", bool(torch.equal(o0, o1) and torch.equal(l0, l1)))
