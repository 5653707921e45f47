import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import torch
from llm_quest_amd import kernels as K
from test_kernels_gpu import _attn_ref
BF16 = torch.bfloat16
def case(B, S, Hq, Hkv, causal, mask):
    D = 128
    g = torch.Generator().manual_seed(S * 7 + Hq)
    q = torch.randn(B * S, Hq * D, generator=g).to(BF16); k = torch.randn(B * S, Hkv * D, generator=g).to(BF16); v = torch.randn(B * S, Hkv * D, generator=g).to(BF16)
    km = None
    if mask == "holes":
        km = torch.ones(B, S, dtype=torch.uint8); km[:, ::5] = 0; km[1, 0] = 1
    elif mask == "h2":
        km = torch.ones(B, S, dtype=torch.uint8); km[:, 3] = 0
    o_ref, _ = _attn_ref(q, k, v, B, S, Hq, Hkv, D, km, causal)
    o, lse = K.attn_fwd(q.cuda(), k.cuda(), v.cuda(), B, S, Hq, Hkv, D, key_mask=None if km is None else km.cuda(), causal=causal)
    e = (o.float().cpu() - o_ref.float()).view(B, S, Hq, D).norm(dim=-1) / o_ref.float().view(B, S, Hq, D).norm(dim=-1)
    print(B, S, Hq, Hkv, causal, mask, "max err per (b,h):", e.amax(dim=1))
    bad = (e > 1e-2).nonzero()
    print(" bad rows:", bad[:12].tolist(), "count", len(bad))
case(2, 63, 4, 2, False, "holes")
case(2, 63, 4, 2, False, "h2")
case(2, 63, 4, 2, True, "holes")
case(2, 64, 4, 2, False, "holes")
case(2, 200, 4, 2, False, "holes")
case(1, 63, 2, 1, False, "holes")
