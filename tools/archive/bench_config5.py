"""Step time of BASELINE configs[4] (Qwen3.5-style multimodal: Qwen3-ViT 3-D patches on 8 x 224 x 224 frames + hybrid GDN / gated
attention text stack, S = 708 = 512 text tokens + 196 merged vision rows), forward + loss + backward on ONE GPU, plus a per-kernel
HIP-event breakdown of one GDN layer's recurrence kernels.  A parity-test configuration, not the bench line.
GPU box only:  python tools/bench_config5.py [--batch 8] [--steps 5]
8 GPUs (BASELINE's DDP form):  python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 tools/bench_config5.py"""
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
from llm_quest_amd import ddp
rank, world, local = ddp.init_from_env()  # one process per GPU under torch.distributed.run; a plain `python` run is rank 0 of 1
torch.cuda.set_device(local)
dev = f"cuda:{local}"
B = args.batch
cfg = dict(QWEN3_5_08B_CONFIG, img_width=224, img_height=224, context_length=1024)
torch.manual_seed(123)
with torch.device(dev):
    vlm = Qwen3_5VLM(cfg).train()
sync = ddp.sync_for_qwen35(vlm)  # no-op at world size 1
sync.broadcast_parameters([vlm])
torch.manual_seed(123 + rank)
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
    loss = lm.lm_loss(h.reshape(-1, h.shape[-1]), tgt.reshape(-1))
    sync.begin_step()
    loss.backward()
    sync.finish_step()


for _ in range(2):
    step()
def fence():
    if world > 1:
        torch.distributed.barrier()
    torch.cuda.synchronize()


fence()
t0 = time.perf_counter()
for _ in range(args.steps):
    step()
fence()
t = (time.perf_counter() - t0) / args.steps
if world > 1:
    tt = torch.tensor([t], device=dev, dtype=torch.float64)
    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    t = float(tt)
units = world * B * (8 + 512)
if rank == 0:
    print(f"configs[4] Qwen3.5-0.8B VLM fwd+bwd B={B} x {world} GPU: {t*1e3:7.1f} ms/step  {units/t:9.0f} frames+tok/s  {3.492e12*B/t/1e12:6.1f} TFLOP/s per GPU algorithmic (3.492 TF/sample, SURVEY 8d)")
    print(f"peak memory {torch.cuda.max_memory_allocated()/2**30:.1f} GiB")
if world > 1:
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()
