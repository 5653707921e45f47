"""Chunked gated-delta-rule forward (csrc/gdr_chunk.hip) against the sequential kernel and an fp64 recurrence, with timings.  usage: python tools/gdr_chunked_check.py [B] [S]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from llm_quest_amd import kernels_q35 as Q
import exp as X
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 708
H, D = 16, 128
torch.manual_seed(0)
nrm = lambda t: torch.nn.functional.normalize(t.float(), dim=-1)
q = nrm(torch.randn(B * S, H, D, device="cuda")).reshape(B * S, H * D).to(torch.bfloat16)
k = nrm(torch.randn(B * S, H, D, device="cuda")).reshape(B * S, H * D).to(torch.bfloat16)
v = torch.randn(B * S, H * D, device="cuda").to(torch.bfloat16)
beta = torch.sigmoid(torch.randn(B * S, H, device="cuda")).contiguous()
alpha = torch.exp(-torch.exp(torch.randn(H, device="cuda") * 0.5) * torch.nn.functional.softplus(torch.randn(B * S, H, device="cuda"))).contiguous()
o0, _, s0 = Q.gated_delta_rule_fwd(q, k, v, beta, alpha, B, S, H, H, D, D, keep=False, want_state=True)
o1, s1 = X.gated_delta_rule_chunked_fwd(q, k, v, beta, alpha, B, S, H, H, D, D, want_state=True)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
ulp = (o0.view(torch.int16).int() - o1.view(torch.int16).int()).abs()
print(f"chunked vs sequential: out rel l2 {rel(o1, o0):.2e}, max bf16 ulp {int(ulp.max())}, share > 0 ulp {float((ulp > 0).float().mean()):.4f}; state rel l2 {rel(s1, s0):.2e}, max abs {float((s1 - s0).abs().max()):.2e}")
# fp64 recurrence on one (batch row, head)
b_, h_ = B - 1, 3
sl = slice(b_ * S, (b_ + 1) * S)
qq, kk, vv = (t[sl].view(S, H, D)[:, h_].double().cpu() for t in (q, k, v))
be, al = beta[sl, h_].double().cpu(), alpha[sl, h_].double().cpu()
st = torch.zeros(D, D, dtype=torch.float64)
outs = []
for t in range(S):
    st = al[t] * st
    u = be[t] * (vv[t] - st @ kk[t])
    st = st + torch.outer(u, kk[t])
    outs.append(st @ (qq[t] * D ** -0.5))
ref = torch.stack(outs)
print(f"against fp64 (b={b_}, h={h_}): chunked out {rel(o1[sl].view(S, H, D)[:, h_].cpu(), ref):.2e}  sequential out {rel(o0[sl].view(S, H, D)[:, h_].cpu(), ref):.2e};  state chunked {rel(s1[b_, h_].cpu(), st):.2e}  sequential {rel(s0[b_, h_].cpu(), st):.2e}")
def timed(fn, n=20):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
print(f"B={B} S={S}: sequential {timed(lambda: Q.gated_delta_rule_fwd(q, k, v, beta, alpha, B, S, H, H, D, D, keep=False)):.0f} us   chunked {timed(lambda: X.gated_delta_rule_chunked_fwd(q, k, v, beta, alpha, B, S, H, H, D, D)):.0f} us")
