"""Same-process check + A/B of the two dK/dV passes at the headline shape: attn_bwd_dkv_kernel<128, true> (default) against the software-pipelined
attn_bwd_dkv2_kernel (ablation bit 14).  The two must agree bit for bit (dK, dV, and -- through the dQ pass that reads the dS scratch -- d(qkv)).
usage: python tools/ab_dkv2.py [B] [masked]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
masked = len(sys.argv) > 2
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
qkv = r(B * S, (Hq + 2 * Hkv) * D)
qw, kw = (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16), (1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device="cuda").float() / D))
ang = torch.arange(1024, device="cuda").float()[:, None] * inv[None, :]
cos, sin = torch.cat((ang.cos(), ang.cos()), -1).contiguous(), torch.cat((ang.sin(), ang.sin()), -1).contiguous()
pos = torch.arange(S, dtype=torch.int32, device="cuda").repeat(B)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
if masked:
    km[::3, S - 150:] = 0
    km[1::5, :40] = 0
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


res = {}
for bit in (0, 16384):
    K._ATTN_ABLATE = bit << 8
    dk.zero_(); dqkv.zero_()
    fused()
    torch.cuda.synchronize()
    res[bit] = (dk.clone(), dqkv.clone())
for name, i in (("dk", 0), ("dqkv (dq | . | dv)", 1)):
    a, b_ = res[16384][i], res[0][i]
    print(f"{name}: equal {torch.equal(a, b_)}  differing {int((a != b_).sum())} of {a.numel()}  rel {float((a.float() - b_.float()).norm() / b_.float().norm()):.3e}  finite {bool(torch.isfinite(a.float()).all())}", flush=True)
for rnd in range(3):
    K._ATTN_ABLATE = 0
    a = timed(fused, 20)
    K._ATTN_ABLATE = 16384 << 8
    b_ = timed(fused, 20)
    print(f"round {rnd}: default {a:7.1f} us   pipelined {b_:7.1f} us per layer (delta + dK/dV + dQ)", flush=True)
