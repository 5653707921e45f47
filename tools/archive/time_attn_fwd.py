"""Forward attention time at the headline shape (and the ViT shape).  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
for (B, S, HQ, HKV, D, causal) in ((64, 708, 16, 8, 128, True), (64, 197, 12, 12, 64, False)):
    g = torch.Generator().manual_seed(3)
    mk = lambda w: torch.randn(B * S, w * D, generator=g).to(torch.bfloat16).cuda()
    q, k, v = mk(HQ), mk(HKV), mk(HKV)
    for _ in range(3): o, lse = K.attn_fwd(q, k, v, B, S, HQ, HKV, D, causal=causal)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): o, lse = K.attn_fwd(q, k, v, B, S, HQ, HKV, D, causal=causal)
    e1.record(); torch.cuda.synchronize()
    print(f"attn_fwd B={B} S={S} Hq={HQ} D={D} causal={causal}: {e0.elapsed_time(e1) / 20 * 1e3:8.1f} us", flush=True)
