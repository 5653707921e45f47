"""QK-norm + RoPE backward at the headline shape: all heads against the key heads only (dq = None), for a list of grid sizes.  usage: python tools/time_qknorm_bwd.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
B, S, Hq, Hkv, D = 64, 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
qkv = r(B * S, (Hq + 2 * Hkv) * D)
qw, kw = r(D), r(D)
inv = 1.0 / (1e6 ** (torch.arange(0, D, 2, device="cuda").float() / D))
ang = torch.arange(1024, device="cuda").float()[:, None] * inv[None, :]
cos, sin = torch.cat((ang.cos(), ang.cos()), -1).contiguous(), torch.cat((ang.sin(), ang.sin()), -1).contiguous()
pos = torch.arange(S, dtype=torch.int32, device="cuda").repeat(B)
q, k, rstd = K.qknorm_rope_fwd(qkv, qw, kw, cos, sin, pos, Hq, Hkv, D)
dq, dk, dqkv = r(B * S, Hq * D), r(B * S, Hkv * D), torch.empty_like(qkv)
def timed(fn, n=50):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for parts in (512, 768, 1024, 1280, 2048, 4096):
    K.QK_PARTS = parts
    a = timed(lambda: K.qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, dq, dk, dqkv, Hq, Hkv, D))
    b = timed(lambda: K.qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, None, dk, dqkv, Hq, Hkv, D))
    print(f"blocks {parts:5d}: all heads {a:6.1f} us   key heads only {b:6.1f} us (incl. the reduce launch)", flush=True)
