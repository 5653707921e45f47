import torch
a = torch.randn(32768, 1024, device="cuda").bfloat16(); b = torch.randn(151936, 1024, device="cuda").bfloat16()
out = torch.empty(32768, 151936, device="cuda", dtype=torch.bfloat16)
for _ in range(4): torch.matmul(a, b.t(), out=out)
x = torch.randn(45376, 1024, device="cuda").bfloat16(); w = torch.randn(4096, 1024, device="cuda").bfloat16()
for _ in range(4): torch.matmul(x, w.t())
torch.cuda.synchronize()
