"""Cross-entropy pass at the headline shape (32 832 rows x 151 936, loss + in-place gradient).  MI355_CE_ROW_IN_REGISTERS=0 selects the two-read kernel.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K
rows, V = 32832, 151936
g = torch.Generator().manual_seed(1)
base = (torch.randn(1024, V, generator=g) * 2).to(torch.bfloat16).cuda()
logits = base.repeat(rows // 1024 + 1, 1)[:rows].contiguous()
tg = torch.randint(0, V, (rows,), generator=g).cuda()
tg[::7] = -100
sc = torch.tensor([1.0 / rows], dtype=torch.float32, device="cuda")
work = logits.clone()
loss_rows, _ = K.cross_entropy(work, tg, want_grad=True, grad_scale=sc)
ref = torch.nn.functional.cross_entropy(logits[:2048].float(), tg[:2048], ignore_index=-100, reduction="none")
print("max |loss - torch fp32| over 2048 rows:", float((loss_rows[:2048] - ref).abs().max()))
lg = logits[:64].float().requires_grad_(True)
torch.nn.functional.cross_entropy(lg, tg[:64], ignore_index=-100, reduction="sum").backward()
print("grad rel l2 vs torch fp32 (64 rows):", float((work[:64].float() - lg.grad * float(sc)).norm() / (lg.grad * float(sc)).norm()))
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for _ in range(2):
    work.copy_(logits); K.cross_entropy(work, tg, want_grad=True, grad_scale=sc)
ts = []
for _ in range(5):
    work.copy_(logits); torch.cuda.synchronize()
    e0.record(); K.cross_entropy(work, tg, want_grad=True, grad_scale=sc); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1))
print(f"cross entropy {rows} x {V}: {min(ts):.3f} ms (min of 5), {sum(ts)/len(ts):.3f} mean; 20 GB algorithmic -> {2*rows*V*2/min(ts)/1e9:.2f} TB/s")
