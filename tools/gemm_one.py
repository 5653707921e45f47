"""Run one GEMM shape a few times (for rocprofv3 --pmc):  python tools/gemm_one.py FORM TILE M N K [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K
form = {"NT": L.GEMM_NT, "NN": L.GEMM_NN, "TN": L.GEMM_TN}[sys.argv[1]]
tile, M, N, Kd = (int(x) for x in sys.argv[2:6])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 3
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
if form == L.GEMM_NT: a, b = r(M, Kd), r(N, Kd)
elif form == L.GEMM_NN: a, b = r(M, Kd), r(Kd, N)
else: a, b = r(Kd, M), r(Kd, N)
out = K.gemm(form, a, b, tile=tile, allow_split_k=False)
for _ in range(reps): K.gemm(form, a, b, out=out, tile=tile, allow_split_k=False)
torch.cuda.synchronize()
