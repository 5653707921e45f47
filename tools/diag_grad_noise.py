"""Diagnostic: per-parameter gradient distance to the CPU oracle's fp32 twin, next to the oracle's own bf16 floor, for every parameter of chosen
decoder layers at full model size (the arithmetic of tests/test_fullsize_properties_gpu.py::test_full_size_step_matches_the_cpu_oracle).
usage: python tools/diag_grad_noise.py [layers, e.g. 20,27]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from llm_quest_amd.multimodal.vlm_engine import vlm_step_loss
from oracle import models

layers = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "20,27").split(",")]
BF16, F32 = torch.bfloat16, torch.float32
torch.set_num_threads(min(16, bench.usable_cores()))
dev = torch.device("cuda", 0)
vit, vit_cfg, ad, llm, llm_cfg = bench.build_models(dev)
img, ids, mask = bench.synthetic_batch(2, "cpu", seed=11, ragged=True)
from llm_quest_amd import kernels as K

names = [n for n, _ in llm.named_parameters() if any(n.startswith(f"trf_blocks.{l}.") for l in layers)]
named = dict(llm.named_parameters())


vit_sd0 = {k: v.detach().cpu() for k, v in vit.state_dict().items()}
with torch.no_grad():
    hid_oracle = models.vit_forward(vit_sd0, vit_cfg, img, output_hidden_states=True).to(dev)  # the reference's fp32 tower


def gpu_run(spill, ablate, oracle_tower=False):
    K._ATTN_DS_SPILL, K._ATTN_ABLATE = spill, ablate << 8
    llm.zero_grad(set_to_none=True)
    ad.zero_grad(set_to_none=True)
    vlm_step_loss(vit, llm, ad, img.to(dev), ids.to(dev), mask.to(dev), hf_vit_model=False, vit_hidden=hid_oracle if oracle_tower else None).backward()
    return {n: named[n].grad.float().cpu() for n in names}


settings = {"default (lean fwd: c folded into Q; lean dK/dV: c folded into K)": (True, 0), "default kernels, vision hidden states from the oracle's fp32 tower": (True, 0, True)}
runs = {k: gpu_run(*v) for k, v in settings.items()}
vit_sd = {k: v.detach().cpu() for k, v in vit.state_dict().items()}
skip = ("mask", "cos", "sin", "out_head.weight")


def oracle_run(dtype):
    ad_sd = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in ad.state_dict().items()}
    llm_sd = {k: v.detach().cpu().to(dtype).requires_grad_(True) for k, v in llm.state_dict().items() if k not in skip}
    llm_sd["out_head.weight"] = llm_sd["emb_dict.weight"]
    _, logits, _ = models.vlm_forward_loss(vit_sd, vit_cfg, ad_sd, llm_sd, dict(llm_cfg, dtype=dtype), img, ids, mask)
    l32 = torch.nn.functional.cross_entropy(logits.float()[:, 196:-1].flatten(0, 1), ids.masked_fill(~mask, -100).flatten(), ignore_index=-100)
    l32.backward()
    return {n: llm_sd[n].grad.float() for n in names}


lo, hi = oracle_run(BF16), oracle_run(F32)
rel = lambda a, b: float((a.double() - b.double()).norm() / b.double().norm())
for tag, mine in runs.items():
    print("==", tag)
    for n in names:
        f, m = rel(lo[n], hi[n]), rel(mine[n], hi[n])
        print(f"{n:48s} mine {m:.3e}  floor {f:.3e}  ratio {m / f:5.2f}")
