"""The block's grouped weight-gradient launch and the LM head's weight gradient (TN, tile 5), isolated; correctness against an fp32 reference on a slice.  GPU box only."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from llm_quest_amd import _lib as L, kernels as K
r = lambda *s: (0.1 * torch.randn(*s, device="cuda")).to(torch.bfloat16)
M = int(sys.argv[1]) if len(sys.argv) > 1 else 113440
def timed(fn, n=6):
    for _ in range(2): fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
x, ctx, h = r(M, 1024), r(M, 2048), r(M, 3072)
dqkv, dy, dgu = r(M, 4096), r(M, 1024), r(M, 6144)
shapes = [(dqkv, x), (dy, ctx), (dgu, x), (dy, h)]
ps = [(a, b, torch.empty(a.shape[1], b.shape[1], dtype=torch.bfloat16, device="cuda"), None) for a, b in shapes]
K.gemm_grouped(L.GEMM_TN, ps)
ref = (dy[:20000].float().t() @ ctx[:20000].float())
got = torch.empty(1024, 2048, dtype=torch.float32, device="cuda")
K.gemm(L.GEMM_TN, dy[:20000], ctx[:20000], out=got, out_dtype=torch.float32, tile=5, allow_split_k=False)
print("rel l2 vs fp32 reference (20 000-row slice, fp32 out):", float((got - ref).norm() / ref.norm()), flush=True)
flop = sum(2.0 * M * a.shape[1] * b.shape[1] for a, b in shapes)
for rep in range(3):
    t = timed(lambda: K.gemm_grouped(L.GEMM_TN, ps))
    print(f"grouped block wgrads: {t:8.1f} us  {flop / t / 1e6:7.1f} TFLOP/s", flush=True)
dl, hh = r(81920, 151936 // 8), r(81920, 1024)
o = torch.empty(151936 // 8, 1024, dtype=torch.bfloat16, device="cuda")
for rep in range(2):
    t = timed(lambda: K.gemm(L.GEMM_TN, dl, hh, out=o, tile=5, allow_split_k=False))
    print(f"LM-head wgrad (1/8 of the vocabulary): {t:8.1f} us  {2.0 * 81920 * (151936 // 8) * 1024 / t / 1e6:7.1f} TFLOP/s", flush=True)
