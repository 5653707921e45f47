"""The NT projections of the step at batch 160 on the default tile, isolated (plain, SwiGLU-forward, SwiGLU-backward, residual, delta epilogues).  Under a
-DGEMM_EPI_ABL=1 build (tools/build_variant.sh epi1 gemm_p2 ...) the write-out is cut away: the difference is what prologue-free overlap could at most buy.
GPU box only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 113440


def timed(fn, n=10):
    for _ in range(3): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


x1 = r(M, 1024)
cases = []
w = r(4096, 1024); cases.append(("QKV fwd N=4096 K=1024", lambda: K.gemm(L.GEMM_NT, x1, w), 2.0 * M * 4096 * 1024))
wgu = r(6144, 1024); cases.append(("gate-up + SwiGLU fwd N=6144 K=1024", lambda: K.gemm_gateup_swiglu(x1, wgu), 2.0 * M * 6144 * 1024))
w2 = r(1024, 3072); gu = r(M, 6144); cases.append(("down dgrad + SwiGLU bwd N=3072 K=1024", lambda: K.gemm_dgrad_swiglu_bwd(x1, w2, gu), 2.0 * M * 3072 * 1024))
wo_t = r(2048, 1024); cases.append(("dctx dgrad N=2048 K=1024 (plain)", lambda: K.gemm(L.GEMM_NT, x1, wo_t), 2.0 * M * 2048 * 1024))
ctx = r(M, 2048); wo = r(1024, 2048); cases.append(("out-proj fwd + residual N=1024 K=2048", lambda: K.gemm(L.GEMM_NT, ctx, wo, residual=x1), 2.0 * M * 1024 * 2048))
act = r(M, 3072); wd = r(1024, 3072); cases.append(("down fwd + residual N=1024 K=3072", lambda: K.gemm(L.GEMM_NT, act, wd, residual=x1), 2.0 * M * 1024 * 3072))
dq = r(M, 4096); wq_t = r(1024, 4096); cases.append(("dqkv dgrad N=1024 K=4096", lambda: K.gemm(L.GEMM_NT, dq, wq_t), 2.0 * M * 1024 * 4096))
for name, fn, flop in cases:
    ts = [timed(fn) for _ in range(2)]
    print(f"{os.environ.get('MI355_LIB_PATH', 'default')[-20:]:20s} {name:42s} " + " / ".join(f"{t:8.1f} us {flop / t / 1e6:7.1f} TF" for t in ts), flush=True)
