"""Which kernels the vendor library picks for the step's GEMM shapes (run under rocprofv3 --kernel-trace --stats; yardstick only)."""
import torch
T = 45376
for name, N, Kd in [("qkv", 4096, 1024), ("gate_up", 6144, 1024), ("down", 1024, 3072)]:
    x = torch.randn(T, Kd, device="cuda").bfloat16(); w = torch.randn(N, Kd, device="cuda").bfloat16(); dy = torch.randn(T, N, device="cuda").bfloat16()
    for _ in range(5):
        torch.matmul(x, w.t()); torch.matmul(dy, w); torch.matmul(dy.t(), x)
torch.cuda.synchronize()
