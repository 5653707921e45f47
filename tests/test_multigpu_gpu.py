"""N > 1 on real hardware: two RCCL ranks (one per GPU) run the composed VLM step on different halves of a batch; the averaged gradients
must equal the single-process gradients of the whole batch, with ragged text masks (token-weighted exchange).  Needs >= 2 GPUs and
skips on the one-GPU test boxes -- there the same protocol is covered by tests/test_ddp_cpu.py (gloo, world 2) and by the one-rank
RCCL test in tests/test_models_gpu.py.  ``bench.py --gpus N`` is checked to refuse a box with fewer than N GPUs before any rendezvous."""

import os
import socket
import subprocess
import sys

import pytest
import torch

from conftest import ROOT, sub_dict
from oracle.gen_golden import TINY_QWEN, TINY_VIT

pytestmark = pytest.mark.gpu
BF16 = torch.bfloat16


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _models(t, dev):
    from llm_quest_amd.multimodal.vision_transformer.vit_engine import ViTAdapter
    from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel
    from llm_quest_amd.qwen.qwen3.qwen3_model import Qwen3Model

    vit = ViTModel(dict(TINY_VIT))
    vit.load_state_dict(sub_dict(t, "vit."))
    vit = vit.to(dev).eval()
    for p in vit.parameters():
        p.requires_grad = False
    llm = Qwen3Model(dict(TINY_QWEN))
    llm.load_state_dict(sub_dict(t, "llm."), strict=False)
    ad = ViTAdapter(64, 128, adapter_type="ffn", dtype=BF16)
    ad.load_state_dict(sub_dict(t, "ad."))
    return vit, llm.to(dev).train(), ad.to(dev).train()


def _batch(t):
    """The fixture's two samples, twice, with four different caption lengths (rank 0 gets the long ones)."""
    img = torch.cat([t["in.image"], t["in.image"].flip(0)])
    ids = torch.cat([t["in.ids"], t["in.ids"].flip(1)])
    T = ids.shape[1]
    lengths = torch.tensor([T, T - 1, 2, 3])
    return img, ids, torch.arange(T).unsqueeze(0) < lengths.unsqueeze(1)


def _rank(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    import torch.distributed as dist
    from safetensors.torch import load_file

    from llm_quest_amd import ddp
    from llm_quest_amd.multimodal.vlm_engine import vlm_step_loss

    r, w, local = ddp.init_from_env()
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    t = load_file(os.path.join(ROOT, "tests", "golden", "vlm_tiny.safetensors"))
    vit, llm, ad = _models(t, dev)
    sync = ddp.sync_for_vlm(llm, ad)
    sync.broadcast_parameters([llm, ad, vit])
    img, ids, tm = _batch(t)
    sl = slice(2 * rank, 2 * rank + 2)
    loss = vlm_step_loss(vit, llm, ad, img[sl].to(dev), ids[sl].to(dev), tm[sl].to(dev), hf_vit_model=False)
    sync.begin_step()
    (loss * sync.loss_weight(tm[sl].sum().to(dev))).backward()
    sync.finish_step()
    torch.cuda.synchronize()
    if rank == 0:
        q.put({n: p.grad.float().cpu() for n, p in list(llm.named_parameters()) + list(ad.named_parameters())})
    dist.barrier()
    dist.destroy_process_group()


def test_two_rccl_ranks_equal_the_single_process_global_batch(golden):
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        pytest.skip("needs two MI355X (the driver's 8-GPU node); one-GPU boxes cover the protocol with gloo + a one-rank RCCL group")
    import torch.multiprocessing as mp

    from llm_quest_amd.multimodal.vlm_engine import vlm_step_loss

    t = golden("vlm_tiny")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    vit, llm, ad = _models(t, torch.device("cuda", 0))
    img, ids, tm = _batch(t)
    vlm_step_loss(vit, llm, ad, img.cuda(), ids.cuda(), tm.cuda(), hf_vit_model=False).backward()
    for n, p in list(llm.named_parameters()) + list(ad.named_parameters()):
        ref = p.grad.float().cpu().double()
        err = float((got[n].double() - ref).norm())
        # two bf16-rounded half-batch gradients averaged vs one bf16-rounded whole-batch gradient: bf16 resolution, not bit equality
        assert err <= 2e-2 * float(ref.norm()) + 1e-6, (n, err, float(ref.norm()))


def test_bench_refuses_more_ranks_than_gpus():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    n = torch.cuda.device_count() + 1
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True, timeout=120)
    assert res.returncode != 0 and "GPU(s) are visible" in (res.stderr + res.stdout)
    env = dict(os.environ, WORLD_SIZE="3", RANK="0")
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert res.returncode != 0 and "WORLD_SIZE=3" in (res.stderr + res.stdout)


@pytest.mark.timeout(900)
def test_bench_two_rank_rehearsal_runs_the_whole_n_gt_1_path_on_one_gpu():
    """``MI355_DDP_REHEARSAL=1 python bench.py --gpus 2`` as a fresh child process on a one-GPU box: the bench starts its own two ranks
    (torch.distributed.run), they share the GPU and exchange over gloo -- rank start-up, parameter broadcast, every block bucket on the
    communication stream, the tied head / embedding bucket (dense or in two parts, by bytes), barriers, max-over-ranks timing and the ONE JSON
    line all execute; only RCCL itself does not.  The line must say it is a rehearsal, and rank 0's loss must equal the one-rank run's
    (same weights by broadcast, rank 0's shard = the one-rank batch: seeds 123 + rank)."""
    import json

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    env = dict(os.environ, MI355_DDP_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    common = ["--batch", "4", "--steps", "2", "--warmup", "1", "--cpu-baseline", "off", "--optimizer", "off"]
    two = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"] + common, env=env, capture_output=True, text=True, timeout=600)
    assert two.returncode == 0, two.stderr[-3000:]
    lines = [ln for ln in two.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, two.stdout[-2000:]
    d2 = json.loads(lines[0])
    assert d2["rehearsal"] is True and d2["n_gpus"] == 2 and d2["rccl_ranks"] == 0 and d2["config"]["global_batch"] == 8 and d2["scaling"] == "weak"
    assert d2["value"] > 0 and d2["steps"] == 2
    # bucket groups + GEMM windows (ddp.GradSync): only the two launches behind each group (and behind the early dense part of the tied bucket) leave the persistent
    # kernel -- not the whole backward.  At batch 4 no launch is persistent-sized: a step's windows never count down, merge into one and end in finish_step.
    gw = d2["gemm_windows"]
    # (window_launches starts at 2 and is then sized from the measured collective / launch times: ddp.GradSync._retune_window)
    assert gw["bucket_blocks"] == 7 and 1 <= gw["window_launches"] <= 16 and gw["windows"] >= 3 and gw["inside_window"] == 0, gw
    env1 = {k: v for k, v in os.environ.items() if k != "MI355_DDP_REHEARSAL"}
    one = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1"] + common, env=env1, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-3000:]
    d1 = json.loads([ln for ln in one.stdout.splitlines() if ln.startswith("{")][0])
    assert "rehearsal" not in d1 and d1["n_gpus"] == 1
    assert d1["loss"] == d2["loss"], (d1["loss"], d2["loss"])
