"""qknorm_rope_bwd time vs relative placement of its buffers (HBM channel aliasing check)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
BF16 = torch.bfloat16
B, S, Hq, Hkv, D = 64, 709, 16, 8, 128
T = B * S
pool = torch.empty(4 * 1024**3, dtype=torch.uint8, device="cuda")
def carve(off, rows, cols):
    n = rows * cols * 2
    return pool[off:off + n].view(BF16).view(rows, cols), off + n
cos = torch.randn(1024, D, device="cuda"); sin = torch.randn(1024, D, device="cuda")
pos = (torch.arange(T, device="cuda") % S).int()
qw = torch.ones(D, device="cuda", dtype=BF16); kw = torch.ones(D, device="cuda", dtype=BF16)
rstd = torch.rand(T, Hq + Hkv, device="cuda")
for pad in (0, 4096):
    off = 0
    qkv, off = carve(off, T, (Hq + 2 * Hkv) * D); off += pad
    dq, off = carve(off, T, Hq * D); off += pad
    dk, off = carve(off, T, Hkv * D); off += pad
    dqkv, off = carve(off, T, (Hq + 2 * Hkv) * D)
    qkv.normal_(); dq.normal_(); dk.normal_()
    for _ in range(3): K.qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, dq, dk, dqkv, Hq, Hkv, D)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10): K.qknorm_rope_bwd(qkv, qw, kw, cos, sin, pos, rstd, dq, dk, dqkv, Hq, Hkv, D)
    e.record(); torch.cuda.synchronize()
    print(f"pad {pad:9d}: {s.elapsed_time(e) * 100:.1f} us   bases mod 2MiB: {[x.data_ptr() % (2 << 20) for x in (qkv, dq, dk, dqkv)]}", flush=True)
