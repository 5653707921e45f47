"""Ablation switches of the alternating-group GEMM loop on one full-occupancy shape per form (GPU box only)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
M = N = 4096; Kd = 8192
TILE = int(os.environ.get("GEMM_TILE", "3"))
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
_w = r(4096, 4096)
for _ in range(200): _w @ _w  # clocks up before anything is timed
torch.cuda.synchronize()
for name, form, sa, sb in (("NT", L.GEMM_NT, (M, Kd), (N, Kd)), ("NN", L.GEMM_NN, (M, Kd), (Kd, N)), ("TN", L.GEMM_TN, (Kd, M), (Kd, N))):
    a, b = r(*sa), r(*sb)
    res = []
    for ab in [int(x, 0) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,1,2,3,4,8").split(",")]:
        tile = TILE | (ab << 8)
        out = K.gemm(form, a, b, tile=tile, allow_split_k=False)
        for _ in range(3): K.gemm(form, a, b, out=out, tile=tile, allow_split_k=False)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): K.gemm(form, a, b, out=out, tile=tile, allow_split_k=False)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        res.append(f"ab{ab}: {2.0*M*N*Kd/ms/1e9:6.0f}")
    print(name, " | ".join(res), flush=True)
