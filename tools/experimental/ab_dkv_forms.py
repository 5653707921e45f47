"""Same-process A/B at the headline shape of the attention backward as the step runs it (dQ pass ending in the query-norm backward), the dK/dV pass in several forms
selected by ablation bits: 0 = persistent workgroup per (batch, kv head) pair; 65536 = one workgroup per key block; 16384 = pipelined kernel (persistent); 81920 = pipelined,
one workgroup per key block.  Gradients are compared with form 0.  usage: python tools/ab_dkv_forms.py [B]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
forms = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 65536, 16384, 81920]
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
qkv = r(B * S, (Hq + 2 * Hkv) * D)
qw, kw = (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16), (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device="cuda").float() / D))
ang = torch.arange(1024, device="cuda").float()[:, None] * inv[None, :]
cos, sin = torch.cat((ang.cos(), ang.cos()), -1).contiguous(), torch.cat((ang.sin(), ang.sin()), -1).contiguous()
pos = torch.arange(S, dtype=torch.int32, device="cuda").repeat(B)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
if len(sys.argv) <= 3:
    km[::3, S - 150:] = 0
q, k, rstd = K.qknorm_rope_fwd(qkv, qw, kw, cos, sin, pos, Hq, Hkv, D)
v = qkv[:, (Hq + Hkv) * D:]
o, lse = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
do = r(B * S, Hq * D)
dk, dqkv = torch.empty_like(k), torch.empty_like(qkv)


def fused():
    K.attn_bwd_qnorm(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dk, dqkv[:, (Hq + Hkv) * D:], qkv, qw, cos, sin, pos, rstd, dqkv, key_mask=km, causal=True)


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


ref = None
for f in forms:
    K._ATTN_ABLATE = f << 8
    dk.zero_(); dqkv.zero_()
    fused()
    torch.cuda.synchronize()
    if ref is None:
        ref = (dk.clone(), dqkv.clone())
    else:
        print(f"form {f}: dk equal {torch.equal(dk, ref[0])}, dqkv equal {torch.equal(dqkv, ref[1])}", flush=True)
for rnd in range(3):
    line = []
    for f in forms:
        K._ATTN_ABLATE = f << 8
        line.append(f"{f}: {timed(fused, 15):7.1f}")
    print(f"round {rnd} (us per layer, delta + dK/dV + dQ): " + "   ".join(line), flush=True)
