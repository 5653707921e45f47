"""Step time of BASELINE configs[4] (Qwen3.5-style multimodal: Qwen3-ViT 3-D patches on 8 x 224 x 224 frames + hybrid GDN / gated
attention text stack, S = 708 = 512 text tokens + 196 merged vision rows), forward + loss + backward on ONE GPU, plus a per-kernel
HIP-event breakdown of one GDN layer's recurrence kernels.  A parity-test configuration, not the bench line.
GPU box only:  python tools/bench_config5.py [--batch 8] [--steps 5]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd.config import QWEN3_5_08B_CONFIG
from llm_quest_amd.qwen.qwen3_5.qwen3_5_vlm_model import Qwen3_5VLM

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=8)
ap.add_argument("--steps", type=int, default=5)
ap.add_argument("--text-only", action="store_true")
args = ap.parse_args()
dev = "cuda"
B = args.batch
cfg = dict(QWEN3_5_08B_CONFIG, img_width=224, img_height=224, context_length=1024)
torch.manual_seed(123)
with torch.device(dev):
    vlm = Qwen3_5VLM(cfg).train()
n_img = (8 // 2) * (14 // 2) * (14 // 2)  # 196 merged rows
ids = torch.randint(0, 248_000, (B, 512 + n_img), device=dev)
ids[:, 100 : 100 + n_img] = cfg["image_token_id"]
pix = torch.randn(B, 3, 8, 224, 224, device=dev)
tgt = torch.roll(ids, -1, 1)
lm = vlm.language_model


def step():
    vlm.zero_grad(set_to_none=True)
    emb = lm.emb_dict(ids)
    if args.text_only:
        pos = None
    else:
        from llm_quest_amd.qwen.qwen3_5.qwen3_5_vlm_model import fuse_vision_embeddings
        vis = vlm.vision_model(pix)
        mask = ids == cfg["image_token_id"]
        emb = fuse_vision_embeddings(emb, mask, vis)
        pos = vlm.compute_3d_position_ids(ids, vlm.get_feeds_3d_shape(pix), image_mask=mask)
    h = lm.forward_hidden(inputs_embs=emb, position_ids=pos)
    lm.lm_loss(h.reshape(-1, h.shape[-1]), tgt.reshape(-1)).backward()


for _ in range(2):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
torch.cuda.synchronize()
t = (time.perf_counter() - t0) / args.steps
units = B * (8 + 512)
print(f"configs[4] Qwen3.5-0.8B VLM fwd+bwd B={B}: {t*1e3:7.1f} ms/step  {units/t:9.0f} frames+tok/s  {3.492e12*B/t/1e12:6.1f} TFLOP/s algorithmic (3.492 TF/sample, SURVEY 8d)")
print(f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
