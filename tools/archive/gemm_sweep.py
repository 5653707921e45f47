"""Measure every GEMM tile configuration on the shapes of the VLM step (HIP events, random data).  GPU box only:
    python tools/gemm_sweep.py [--batch 32]"""
import argparse, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import _lib as L, kernels as K

ap = argparse.ArgumentParser(); ap.add_argument("--batch", type=int, default=32); ap.add_argument("--tiles", default="1,2,3"); ap.add_argument("--only", default=""); ap.add_argument("--lib", action="store_true"); args = ap.parse_args()
TILES = [int(x) for x in args.tiles.split(",")]
M, Mh = args.batch * 709, args.batch * 512
dev = "cuda"
rnd = lambda *s: torch.randn(*s, device=dev).to(torch.bfloat16)
shapes = [  # name, form, A shape, B shape, flops
    ("qkv fwd NT", L.GEMM_NT, (M, 1024), (4096, 1024)), ("out fwd NT", L.GEMM_NT, (M, 2048), (1024, 2048)),
    ("gateup fwd NT", L.GEMM_NT, (M, 1024), (6144, 1024)), ("down fwd NT", L.GEMM_NT, (M, 3072), (1024, 3072)),
    ("head fwd NT", L.GEMM_NT, (Mh, 1024), (151936, 1024)),
    ("d_act NN (N=3072)", L.GEMM_NN, (M, 1024), (1024, 3072)), ("d_h2 NN (K=6144)", L.GEMM_NN, (M, 6144), (6144, 1024)),
    ("d_ctx NN (N=2048)", L.GEMM_NN, (M, 1024), (1024, 2048)), ("d_h1 NN (K=4096)", L.GEMM_NN, (M, 4096), (4096, 1024)),
    ("head dgrad NN", L.GEMM_NN, (Mh, 151936), (151936, 1024)),
    ("dW gateup TN", L.GEMM_TN, (M, 6144), (M, 1024)), ("dW down TN", L.GEMM_TN, (M, 1024), (M, 3072)),
    ("dW qkv TN", L.GEMM_TN, (M, 4096), (M, 1024)), ("dW out TN", L.GEMM_TN, (M, 1024), (M, 2048)),
    ("dW head TN", L.GEMM_TN, (Mh, 151936), (Mh, 1024)),
]
for name, form, sa, sb in shapes:
    if args.only and args.only not in name: continue
    a, b = rnd(*sa), rnd(*sb)
    if form == L.GEMM_NT: flops = 2.0 * sa[0] * sb[0] * sa[1]
    elif form == L.GEMM_NN: flops = 2.0 * sa[0] * sb[1] * sa[1]
    else: flops = 2.0 * sa[1] * sb[1] * sa[0]
    res = []
    for tile in TILES:
        out = K.gemm(form, a, b, tile=tile)
        for _ in range(2): K.gemm(form, a, b, out=out, tile=tile)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): K.gemm(form, a, b, out=out, tile=tile)
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        res.append(f"t{tile:#x}: {ms*1e3:7.1f} us {flops/ms/1e9:7.1f} TF")
    if args.lib:  # calibration only: what the vendor library reaches on the same operands
        fn = {L.GEMM_NT: lambda: a @ b.t(), L.GEMM_NN: lambda: a @ b, L.GEMM_TN: lambda: a.t() @ b}[form]
        for _ in range(3): o2 = fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(5): o2 = fn()
        e.record(); torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 5
        res.append(f"lib: {ms*1e3:7.1f} us {flops/ms/1e9:7.1f} TF"); del o2
    print(f"{name:22s} " + " | ".join(res), flush=True)
    del a, b, out
