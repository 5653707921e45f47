import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from llm_quest_amd import kernels_q35 as Q
import exp as X
B, S, H, D = 8, 708, 16, 128
q = torch.randn(B * S, H * D, device="cuda").to(torch.bfloat16); k = torch.randn_like(q); v = torch.randn_like(q)
beta = torch.rand(B * S, H, device="cuda"); alpha = 0.5 + 0.5 * torch.rand(B * S, H, device="cuda")
def timed(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(os.environ.get("MI355_GDR_ABL", "0"), f"{timed(lambda: X.gated_delta_rule_chunked_fwd(q, k, v, beta, alpha, B, S, H, H, D, D)):.0f} us")
