"""Same-process A/B at the headline shape: attention backward + QK-norm / RoPE backward as a pair of launches against the form whose dQ pass ends
in the query heads' norm backward (mi355_attn_bwd_qnorm + the key heads' kernel).  usage: python tools/time_attn_bwd_qnorm.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
qkv = r(B * S, (Hq + 2 * Hkv) * D)
qw, kw = (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16), (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device="cuda").float() / D))
ang = torch.arange(1024, device="cuda").float()[:, None] * inv[None, :]
cos, sin = torch.cat((ang.cos(), ang.cos()), -1).contiguous(), torch.cat((ang.sin(), ang.sin()), -1).contiguous()
pos = torch.arange(S, dtype=torch.int32, device="cuda").repeat(B)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
q, k, rstd = K.qknorm_rope_fwd(qkv, qw, kw, cos, sin, pos, Hq, Hkv, D)
v = qkv[:, (Hq + Hkv) * D:]
o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
do = r(B * S, Hq * D)
dq, dk, dqkv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(qkv)


def pair():
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dqkv[:, (Hq + Hkv) * D:], key_mask=km, causal=True)
    K.qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, dq, dk, dqkv, Hq, Hkv, D)


def fused():
    K.attn_bwd_qnorm(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dk, dqkv[:, (Hq + Hkv) * D:], qkv, qw, cos, sin, pos, rstd, dqkv, key_mask=km, causal=True)
    K.qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, None, dk, dqkv, Hq, Hkv, D)


def timed(fn, n):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for rnd in range(4):
    a, b = timed(pair, 40), timed(fused, 40)
    print(f"round {rnd}: pair {a:7.1f} us   fused {b:7.1f} us per layer", flush=True)
