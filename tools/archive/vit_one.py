"""ViT-B/16 fwd+bwd steps only (for rocprofv3): python tools/vit_one.py [batch] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd.config import VIT_BASE_CONFIG
from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
torch.manual_seed(1)
with torch.device("cuda"):
    vit = ViTModel(dict(VIT_BASE_CONFIG, drop_rate=0.0)).train()
img = torch.randn(B, 3, 224, 224, device="cuda")
y = torch.randint(0, VIT_BASE_CONFIG.get("num_classes", 100), (B,), device="cuda")
for _ in range(steps):
    vit.zero_grad(set_to_none=True)
    torch.nn.functional.cross_entropy(vit(img).float(), y).backward()
torch.cuda.synchronize()
