"""Full-size end-to-end check of the WHOLE train step (BASELINE configs[3] models, per-GPU batch 16): the reference's loop
`vlm_training_loop_simple` (forward, loss, backward, global-norm clip 1.0, optimizer step) on the HIP kernels with ArenaAdamW,
a fixed synthetic batch repeated, loss printed per epoch: it has to fall from ln(V) = 11.93 as the model memorises the batch.
GPU box only:  python tools/train_demo.py [--steps 30] [--batch 16]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from llm_quest_amd.multimodal.vlm_engine import vlm_step_loss, vlm_training_loop_simple
from llm_quest_amd.optim import ArenaAdamW

ap = argparse.ArgumentParser(); ap.add_argument("--steps", type=int, default=30); ap.add_argument("--batch", type=int, default=16)
a = ap.parse_args()
dev = torch.device("cuda", 0)
vit, _, ad, llm, _ = bench.build_models(dev)
img, ids, mask = bench.synthetic_batch(a.batch, dev, seed=123)
batch = {"image": img, "input_ids": ids, "attention_mask": mask}
opt = ArenaAdamW(list(llm.parameters()) + list(ad.parameters()), lr=2e-4, weight_decay=0.0)
opt.attach(llm, ad)
def loss_now():
    with torch.no_grad():
        return float(vlm_step_loss(vit, llm, ad, img, ids, mask, hf_vit_model=False))
print(f"step   0: loss {loss_now():.4f}")
t0 = time.perf_counter()
for s in range(0, a.steps, 5):
    vlm_training_loop_simple(vit, llm, ad, [batch] * 5, opt, 1, dev, hf_vit_model=False, eval_freq=10**9)
    torch.cuda.synchronize()
    print(f"step {s + 5:3d}: loss {loss_now():.4f}   ({(time.perf_counter() - t0) / (s + 5) * 1e3:.0f} ms/step incl. clip + AdamW)")
