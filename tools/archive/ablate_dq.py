"""Time the attention backward (delta + dK/dV + dQ with the query-norm backward as its write-out) at the headline shape under whatever library
MI355_LIB_PATH names (profiling builds of tools/build_variant.sh: -DDQ_ABL=n cuts parts out of the dQ pass, so differences between variants
are the cost of those parts).  usage: MI355_LIB_PATH=build_variants/libmi355vlm_dq4.so python tools/ablate_dq.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
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


ts = [timed(fused, 20) for _ in range(3)]
print(f"{os.environ.get('MI355_LIB_PATH', 'default')}: attention backward (delta + dK/dV + dQ/qnorm) B={B}: " + " / ".join(f"{t:.0f}" for t in ts) + " us", flush=True)

if "dqprof" in os.environ.get("MI355_LIB_PATH", ""):
    import ctypes
    from llm_quest_amd import _lib as L
    lib = L.load()
    out = (ctypes.c_ulonglong * 32)()
    lib.mi355_debug_dq_prof(out, 1)
    fused()
    lib.mi355_debug_dq_prof(out, 0)
    names = ["DMA wait", "tile barrier", "DMA issue", "reads + MFMA", "write-out requests", "wait for slowest wave", "transposition", "arithmetic + stores + dw"]
    for w, o in (("wave 0", 0), ("wave 3", 16)):
        n = max(out[o + 9], 1)
        print(f"{w}: {n} workgroups sampled, {out[o + 10] / n:.1f} tiles each, {out[o + 8] / n:.0f} cycles per workgroup")
        for k, nm in enumerate(names):
            print(f"    {nm:28s} {out[o + k] / n:9.0f} cycles  ({100.0 * out[o + k] / max(out[o + 8], 1):5.1f} %)")
