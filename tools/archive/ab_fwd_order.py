"""Same-process A/B of the attention forward's workgroup order: head by head (ablation bit 8), groups of 16 heads (default), groups of 8 / 32 / 64 (bits 10 / 7 / 6).  usage: python tools/ab_fwd_order.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
q, k, v = r(B * S, Hq * D), r(B * S, Hkv * D), r(B * S, Hkv * D)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
def run(bit, n=100):
    keep = K._ATTN_ABLATE
    K._ATTN_ABLATE = keep | (bit << 8)
    try:
        for _ in range(5): K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n): K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    finally:
        K._ATTN_ABLATE = keep
for rnd in range(4):
    print(f"round {rnd}: head by head {run(256):6.1f} us   groups of 16 {run(0):6.1f} us   groups of 8 {run(1024):6.1f}   32 {run(128):6.1f}   64 {run(64):6.1f} us", flush=True)
