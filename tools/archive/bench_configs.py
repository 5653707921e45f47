"""Step times of the two other single-GPU BASELINE configurations (parity-test cases, not bench lines):
  configs[1]  ViT-Base/16 224x224 forward+backward (classification loss), bf16 MFMA operands over fp32 masters
  configs[2]  Qwen3-0.6B text-only, seq 1024, bf16 forward+backward
GPU box only:  python tools/bench_configs.py [--vit-batch 64] [--llm-batch 8]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd.config import VIT_BASE_CONFIG, qwen3_config_creator
from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel
from llm_quest_amd.qwen.qwen3.qwen3_model import Qwen3Model

ap = argparse.ArgumentParser()
ap.add_argument("--vit-batch", type=int, default=64)
ap.add_argument("--llm-batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=5)
args = ap.parse_args()
dev = "cuda"


def timed(fn, steps):
    for _ in range(2): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


torch.manual_seed(1)
img = torch.randn(args.vit_batch, 3, 224, 224, device=dev)
y = torch.randint(0, VIT_BASE_CONFIG.get("num_classes", 100), (args.vit_batch,), device=dev)
for drop in (0.0, VIT_BASE_CONFIG["drop_rate"]):  # config 2 "with drop_rate=0.1 (as config) and 0.0 (parity)"
    with torch.device(dev):
        vit = ViTModel(dict(VIT_BASE_CONFIG, drop_rate=drop)).train()
    def vit_step():
        vit.zero_grad(set_to_none=True)
        torch.nn.functional.cross_entropy(vit(img).float(), y).backward()
    t = timed(vit_step, args.steps)
    print(f"configs[1] ViT-B/16 fwd+bwd  B={args.vit_batch} drop_rate={drop}: {t*1e3:7.1f} ms/step  {args.vit_batch/t:9.0f} img/s  {105.4e9*args.vit_batch/t/1e12:6.1f} TFLOP/s algorithmic (105.4 GF/img)")
    del vit
    torch.cuda.empty_cache()
del img

with torch.device(dev):
    llm = Qwen3Model(dict(qwen3_config_creator("0.6B"), context_length=1024)).train()
ids = torch.randint(0, 151_936, (args.llm_batch, 1024), device=dev)
def llm_step():
    llm.zero_grad(set_to_none=True)
    h = llm.forward_hidden(ids)
    llm.lm_loss(h.reshape(-1, h.shape[-1]), ids.reshape(-1)).backward()
try:
    t = timed(llm_step, args.steps)
except Exception as e:  # forward_hidden / lm_loss signature differs: fall back to logits + global_loss
    from llm_quest_amd.engine import global_loss
    def llm_step():
        llm.zero_grad(set_to_none=True)
        global_loss(llm(ids), ids).backward()
    t = timed(llm_step, args.steps)
tok = args.llm_batch * 1024
print(f"configs[2] Qwen3-0.6B S=1024 fwd+bwd B={args.llm_batch}: {t*1e3:7.1f} ms/step  {tok/t:9.0f} tok/s  {4.023e12*args.llm_batch/t/1e12:6.1f} TFLOP/s algorithmic (4.023 TF/sample)")
