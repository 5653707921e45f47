"""CALIBRATION ONLY -- never part of the product path: runs bench.py's step with the plain NT projections (no epilogue: QKV, LM-head forward, the NT dgrads without a residual)
handed to the vendor library (torch.matmul -> hipBLASLt), to see what that library's kernels are worth INSIDE the step, where isolated GEMM timings have not ranked loops
correctly before.  usage: python tools/calibrate_lib_in_step.py [which] [bench.py flags]    which = none | qkv | lmhead | all"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K

which = sys.argv[1] if len(sys.argv) > 1 else "all"
sys.argv = [sys.argv[0]] + sys.argv[2:]
orig = K.gemm
count = {"lib": 0, "ours": 0}


def gemm(form, a, b, out=None, out_dtype=K.BF16, bias=None, residual=None, gelu=False, allow_split_k=True, tile=0):
    plain = form == L.GEMM_NT and bias is None and residual is None and not gelu and out_dtype == K.BF16 and a.is_contiguous() and b.is_contiguous()
    N = b.shape[0] if form == L.GEMM_NT else 0
    take = plain and ((which == "all" and N >= 2048) or (which == "qkv" and N == 4096) or (which == "lmhead" and N > 100000))
    if take and (out is None or out.is_contiguous()):
        count["lib"] += 1
        if out is None:
            out = torch.empty((a.shape[0], N), dtype=K.BF16, device=a.device)
        return torch.matmul(a, b.t(), out=out)
    count["ours"] += 1
    return orig(form, a, b, out=out, out_dtype=out_dtype, bias=bias, residual=residual, gelu=gelu, allow_split_k=allow_split_k, tile=tile)


if which != "none":
    K.gemm = gemm
import bench
bench.main()
print(f"calibration run: {count['lib']} GEMM launches went to the vendor library, {count['ours']} stayed", file=sys.stderr)
