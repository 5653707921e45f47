"""CALIBRATION ONLY: this package's attention kernels against torch's scaled_dot_product_attention (the vendor flash-attention path on ROCm) at the headline shape
(B x 16 heads x 709 x 128, causal, the 8 kv heads repeated for the library), forward and forward + backward.  Never a product path.  usage: python tools/attn_vs_lib.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S, Hq, Hkv, D = 709, 16, 8, 128
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
q, k, v, do = r(B * S, Hq * D), r(B * S, Hkv * D), r(B * S, Hkv * D), r(B * S, Hq * D)
km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
ql = q.view(B, S, Hq, D).transpose(1, 2).contiguous().requires_grad_(True)
kl = k.view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1).contiguous().requires_grad_(True)
vl = v.view(B, S, Hkv, D).transpose(1, 2).repeat_interleave(Hq // Hkv, 1).contiguous().requires_grad_(True)
dol = do.view(B, S, Hq, D).transpose(1, 2).contiguous()


def timed(fn, n=30):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def ours_fwd():
    return K.attn_fwd(q, k, v, B, S, Hq, Hkv, D, key_mask=km, causal=True)
def ours_fb():
    o, lse = ours_fwd()
    K.attn_bwd(q, k, v, o, do, lse, B, S, Hq, Hkv, D, dq, dk, dv, key_mask=km, causal=True)
def lib_fwd():
    return F.scaled_dot_product_attention(ql, kl, vl, is_causal=True)
def lib_fb():
    o = lib_fwd()
    torch.autograd.grad(o, (ql, kl, vl), dol)
flop_f = 4 * S * (S + 1) / 2 * D * Hq * B
for backend in ("flash", "efficient", "default"):
    try:
        from torch.nn.attention import sdpa_kernel, SDPBackend
        ctx = sdpa_kernel({"flash": SDPBackend.FLASH_ATTENTION, "efficient": SDPBackend.EFFICIENT_ATTENTION}.get(backend, [SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION, SDPBackend.MATH]))
        with ctx:
            tf, tfb = timed(lib_fwd), timed(lib_fb)
        print(f"library ({backend:9s}): forward {tf:7.1f} us ({flop_f / tf / 1e6:6.1f} TFLOP/s)   forward + backward {tfb:7.1f} us", flush=True)
    except Exception as ex:
        print(f"library ({backend}): not available: {type(ex).__name__}: {str(ex)[:120]}", flush=True)
tf, tfb = timed(ours_fwd), timed(ours_fb)
print(f"this package          : forward {tf:7.1f} us ({flop_f / tf / 1e6:6.1f} TFLOP/s)   forward + backward {tfb:7.1f} us (GQA: K / V read once per kv head, not repeated)")
