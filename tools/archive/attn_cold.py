"""Attention kernels at the headline shape under the conditions of the step rather than of a hot loop: rotating operand sets
(NSETS distinct q/k/v/out sets, together larger than the 256 MB memory-side cache) and the key-padding mask the decoder passes.
usage: python tools/attn_cold.py [B] [NSETS]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
NS = int(sys.argv[2]) if len(sys.argv) > 2 else 4
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
sets = []
for i in range(NS):
    qkv = r(B * S, (Hq + 2 * Hkv) * D)
    sets.append(dict(q=r(B * S, Hq * D), k=r(B * S, Hkv * D), v=qkv[:, (Hq + Hkv) * D:], do=r(B * S, Hq * D), dq=torch.empty(B * S, Hq * D, device="cuda", dtype=torch.bfloat16),
                     dk=torch.empty(B * S, Hkv * D, device="cuda", dtype=torch.bfloat16), dqkv=torch.empty_like(qkv)))
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")


def timed(fn, n):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(0); fn(1 % NS)
    s.record()
    for i in range(n):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for mask in (None, km):
    for rot in (False, True):
        pick = (lambda i: sets[i % NS]) if rot else (lambda i: sets[0])
        def fwd(i):
            s = pick(i)
            s["o"], s["lse"] = K.attn_fwd(s["q"], s["k"], s["v"], B, S, Hq, Hkv, D, key_mask=mask, causal=True)
        for i in range(NS): fwd(i)
        def bwd(i):
            s = pick(i)
            K.attn_bwd(s["q"], s["k"], s["v"], s["o"], s["do"], s["lse"], B, S, Hq, Hkv, D, s["dq"], s["dk"], s["dqkv"][:, (Hq + Hkv) * D:], key_mask=mask, causal=True)
        timed(fwd, 100); tf, tb = timed(fwd, 200), timed(bwd, 100)
        print(f"key_mask={'yes' if mask is not None else 'no '} rotating={'yes' if rot else 'no '}: forward {tf:6.1f} us   backward {tb:6.1f} us", flush=True)
