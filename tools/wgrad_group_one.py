"""The four weight gradients of one Qwen3-0.6B block as the step launches them -- ONE grouped TN launch, 64 + 32 + 96 + 48 = 240 tiles of 256 x 256 on 256 CUs, K = the
per-GPU batch's tokens -- a few times (for rocprofv3 --pmc):  python tools/wgrad_group_one.py [batch] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K

B = int(sys.argv[1]) if len(sys.argv) > 1 else 160
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
M = B * 709
r = lambda *s: torch.randn(*s, device="cuda").to(torch.bfloat16)
shapes = [(4096, 1024), (1024, 2048), (6144, 1024), (1024, 3072)]  # [out, in] of w_queries|w_keys|w_values, out_proj, lin1|lin_gate, lin2
probs = [(r(M, o), r(M, i), torch.zeros(o, i, device="cuda", dtype=torch.bfloat16), None) for o, i in shapes]
for _ in range(reps + 1):
    K.gemm_grouped(L.GEMM_TN, probs)
torch.cuda.synchronize()
