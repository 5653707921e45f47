"""A/B of the attention backward with and without the dS scratch (kernels._ATTN_DS_SPILL) at the headline shape.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K

B, S, HQ, HKV, D = 64, 709, 16, 8, 128
g = torch.Generator().manual_seed(3)
mk = lambda w: torch.randn(B * S, w * D, generator=g).to(torch.bfloat16).cuda()
q, k, v, do = mk(HQ), mk(HKV), mk(HKV), mk(HQ)
o, lse = K.attn_fwd(q, k, v, B, S, HQ, HKV, D, causal=True)
outs = {}
for spill in (False, True, False, True):
    K._ATTN_DS_SPILL = spill
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    for _ in range(3):
        K.attn_bwd(q, k, v, o, do, lse, B, S, HQ, HKV, D, dq, dk, dv, causal=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        K.attn_bwd(q, k, v, o, do, lse, B, S, HQ, HKV, D, dq, dk, dv, causal=True)
    e1.record(); torch.cuda.synchronize()
    print(f"spill={spill}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us per backward", flush=True)
    outs[spill] = (dq.float(), dk.float(), dv.float())
for n, a, b in zip(("dq", "dk", "dv"), outs[False], outs[True]):
    print(n, "rel l2 between the two forms:", float((a - b).norm() / a.norm()), "finite:", bool(torch.isfinite(b).all()))
