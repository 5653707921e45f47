"""Same-process A/B of the attention forward kernels (first generation, lean softmax = the default, the one-wave-per-SIMD experiment) at the headline shape (B x 16 q heads x 8 kv heads, S = 709, head_dim 128, causal):
interleaved rounds, HIP events on torch's current stream (the stream the kernels are launched on).  usage: python tools/time_attn_fwd2.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llm_quest_amd import kernels as K
import exp as X

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, Hq, Hkv, D = 709, 16, 8, 128
g = torch.Generator().manual_seed(0)
qkv = torch.randn(B * S, (Hq + 2 * Hkv) * D, generator=g).to(torch.bfloat16).cuda()
q, k, v = qkv[:, : Hq * D], qkv[:, Hq * D : (Hq + Hkv) * D], qkv[:, (Hq + Hkv) * D :]


def run(bit, n):
    keep = K._ATTN_ABLATE
    fwd = (lambda *a, **kw: X.attn_fwd2(*a, **kw)) if bit == 8192 else K.attn_fwd  # 8192 = the experiment (libmi355exp.so)
    K._ATTN_ABLATE = keep | ((bit & 2048) << 8)
    try:
        for _ in range(3):
            fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(n):
            fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
        e.record()
        torch.cuda.synchronize()
        return s.elapsed_time(e) / n * 1e3
    finally:
        K._ATTN_ABLATE = keep


flop = 4 * S * (S + 1) / 2 * D * Hq * B
for rnd in range(4):
    a, b, c = run(2048, 30), run(0, 30), run(8192, 30)
    print(f"round {rnd}: first-generation {a:7.1f} us ({flop / a / 1e6:6.1f} TFLOP/s)   lean {b:7.1f} us ({flop / b / 1e6:6.1f} TFLOP/s)   one wave per SIMD {c:7.1f} us ({flop / c / 1e6:6.1f} TFLOP/s)")
o1, l1 = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
K._ATTN_ABLATE |= 2048 << 8
o0, l0 = K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, causal=True)
print("rel l2 new vs old:", float((o1.float() - o0.float()).norm() / o0.float().norm()), " max |lse diff|:", float((l1 - l0).abs().max()))
