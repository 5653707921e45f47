"""Frozen ViT-B/16 forward (the tower of the headline step: B = 64, eval, no grad), per NT tile choice.  GPU box only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd import kernels as K, _lib as L
from llm_quest_amd.config import VIT_BASE_CONFIG
from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel
torch.manual_seed(1)
with torch.device("cuda"):
    vit = ViTModel(dict(VIT_BASE_CONFIG, drop_rate=0.0)).eval()
for p_ in vit.parameters(): p_.requires_grad = False
img = torch.randn(64, 3, 224, 224, device="cuda")
def run(n=20):
    with torch.no_grad():
        for _ in range(3): vit(img, output_hidden_states=True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(n): vit(img, output_hidden_states=True)
        e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for name, t in (("auto", 0), ("128x128", 1), ("256x256 tile 2", 2), ("256x256 tile 3", 3), ("auto", 0)):
    K._TILE_BY_FORM[L.GEMM_NT] = t
    print(f"ViT-B/16 forward B=64, NT tile {name:16s}: {run():7.3f} ms", flush=True)
