"""Run attention fwd+bwd at the VLM step's shape (for rocprofv3 / timing):  python tools/attn_one.py [B] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S = int(sys.argv[3]) if len(sys.argv) > 3 else 709
Hq, Hkv, D = 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
qkv = r(B * S, (Hq + 2 * Hkv) * D)
q, k, v = r(B * S, Hq * D), r(B * S, Hkv * D), qkv[:, (Hq + Hkv) * D:]
do = r(B * S, Hq * D)
dq, dk, dqkv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(qkv)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
def run():
    o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dqkv[:, (Hq + Hkv) * D:], key_mask=km, causal=True)
run(); torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(reps): run()
e.record(); torch.cuda.synchronize()
print(f"attn fwd+bwd B={B}: {s.elapsed_time(e)/reps*1e3:.1f} us per layer")
