"""Time the gated-delta-rule kernels alone at config-5 shapes: python tools/gdr_one.py [--batch 8] [--seq 708]"""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels_q35 as Q
ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=8); ap.add_argument("--seq", type=int, default=708); ap.add_argument("--reps", type=int, default=5)
a = ap.parse_args()
B, S, Hqk, Hv, Dk, Dv = a.batch, a.seq, 16, 16, 128, 128
dev = "cuda"
torch.manual_seed(0)
q = torch.nn.functional.normalize(torch.randn(B * S, Hqk, Dk, device=dev), dim=-1).reshape(B * S, -1).bfloat16()
k = torch.nn.functional.normalize(torch.randn(B * S, Hqk, Dk, device=dev), dim=-1).reshape(B * S, -1).bfloat16()
v = torch.randn(B * S, Hv * Dv, device=dev).bfloat16()
beta = torch.rand(B * S, Hv, device=dev); alpha = 0.5 + 0.5 * torch.rand(B * S, Hv, device=dev)
do = torch.randn(B * S, Hv * Dv, device=dev).bfloat16(); dv = torch.empty_like(v)
def t(fn):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / a.reps * 1e3
o, ck, _ = Q.gated_delta_rule_fwd(q, k, v, beta, alpha, B, S, Hqk, Hv, Dk, Dv)
print(f"B={B} S={S}: fwd {t(lambda: Q.gated_delta_rule_fwd(q, k, v, beta, alpha, B, S, Hqk, Hv, Dk, Dv)):8.1f} us   bwd(+reduce) {t(lambda: Q.gated_delta_rule_bwd(q, k, v, beta, alpha, ck, do, dv, B, S, Hqk, Hv, Dk, Dv)):8.1f} us   checkpoint spacing {Q.gdr_chunk()}")
