"""BASELINE config 5 (Qwen3.5 multimodal), the part that is native so far: vision tower + merge adapter forward/backward,
masked-scatter fusion (bit-exact) and 3-D position ids (integer-exact) vs fixtures generated from the reference."""

import numpy as np
import pytest
import torch

from conftest import sub_dict
from oracle.gen_golden import TINY_Q35_VISION

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


def rel_l2(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return float((a - b).norm() / (b.norm() + 1e-30))


def test_patchify3d_and_merge_are_bit_exact(golden):
    from llm_quest_amd import kernels as K
    from oracle import qwen3_5 as q35

    t = golden("qwen35_vision_tiny")
    clip = torch.arange(2 * 3 * 4 * 16 * 16, dtype=F32).view(2, 3, 4, 16, 16)
    rows = K.patchify3d(clip.cuda(), 4, 2, out_dtype=F32).cpu()
    idx = torch.from_numpy(q35.patch3d_gather_index(3, 4, 16, 16, 4, 2))
    assert np.array_equal(idx.numpy(), t["patch3d.gather"].numpy().astype(np.int64))
    n = idx.shape[0]
    for b in range(2):
        assert torch.equal(rows[b * n : (b + 1) * n], clip[b].reshape(-1)[idx])
    # merge permutation and its inverse
    x = torch.randn(3 * 4 * 6, 16).to(BF16)
    merged = K.merge_patches(x.cuda(), 3, 4, 6, 2)
    src = torch.from_numpy(q35.merge_row_source(3, 4, 6, 2))
    assert torch.equal(merged.cpu(), x[src.reshape(-1)].reshape(src.shape[0], 4 * 16))
    assert torch.equal(K.merge_patches(merged, 3, 4, 6, 2, inverse=True).cpu(), x)


def test_masked_scatter_and_position_ids(golden):
    import types

    from llm_quest_amd.qwen.qwen3_5.qwen3_5_vlm_model import Qwen3_5VLM, fuse_vision_embeddings

    t = golden("qwen35_vision_tiny")
    ids = t["pos3d.ids"]
    emb = t["scatter.emb"].cuda().requires_grad_(True)
    vis = t["scatter.vis"].cuda().requires_grad_(True)
    fused = fuse_vision_embeddings(emb, (ids == 999).cuda(), vis)
    assert torch.equal(fused.cpu(), t["scatter.out"])  # bit-exact row-major fill
    g = torch.randn_like(fused)
    fused.backward(g)
    mask = (ids == 999)
    assert torch.equal(emb.grad.cpu()[mask], torch.zeros_like(emb.grad.cpu()[mask]))
    assert torch.equal(emb.grad.cpu()[~mask], g.cpu()[~mask])
    assert vis.grad.dtype == F32 and torch.equal(vis.grad.cpu().reshape(-1, 16), g.cpu()[mask].float())
    stub = types.SimpleNamespace(image_token_id=999, merge_size=2)
    pos = Qwen3_5VLM.compute_3d_position_ids(stub, ids.cuda(), torch.tensor([[2, 4, 4]]))
    assert torch.equal(pos.cpu(), t["pos3d.out"])
    assert torch.equal(Qwen3_5VLM.compute_3d_position_ids(stub, ids.cuda(), None).cpu(), t["pos3d.text_only"])


def test_vision_tower_forward_backward(golden):
    from llm_quest_amd.qwen.qwen3_5.qwen3_5_vision_model import Qwen3_5VisionModel

    t = golden("qwen35_vision_tiny")
    m = Qwen3_5VisionModel(dict(TINY_Q35_VISION))
    assert set(m.state_dict()) == set(sub_dict(t, "vis.sd."))
    m.load_state_dict(sub_dict(t, "vis.sd."))
    m = m.cuda().train()
    out = m(t["vis.in"].cuda())
    assert out.shape == t["vis.out"].shape and out.dtype == F32
    # bf16 MFMA operands over fp32 masters vs the all-fp32 reference tower
    assert rel_l2(out, t["vis.out"]) < 1.5e-2
    out.backward(t["vis.gout"].cuda())
    ref = sub_dict(t, "vis.grad.")
    for name, p in m.named_parameters():
        assert p.grad is not None and p.grad.dtype == F32, name
        err = float((p.grad.double().cpu() - ref[name].double()).norm())
        assert err <= 4e-2 * float(ref[name].double().norm()) + 1e-3, f"{name}: |err| {err:.3e} |ref| {float(ref[name].norm()):.3e}"
    with torch.no_grad():
        assert rel_l2(m.eval()(t["vis.in"].cuda()), t["vis.out"]) < 1.5e-2
