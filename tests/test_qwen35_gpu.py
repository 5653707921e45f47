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


def test_qwen35_vlm_wrapper_forward_backward_vs_reference_fixture(golden):
    """Row a25 against the REFERENCE, not against its own parts: ``Qwen3_5VLM.forward`` (qwen3_5_vlm_model.py:178-227) on the fixture the imported
    reference wrote (oracle/gen_golden.py::gen_qwen35_vlm): fp32 vision tower -> masked_scatter at the placeholders -> 3-D position ids -> bf16
    hybrid text stack, padding mask, logits and EVERY parameter gradient of both towers.  Judged by the reference's own bf16 noise: distance to
    the reference's fp32 weight twin <= 1.5 x the reference-bf16 distance to that twin (no additive slack where that floor is >= 1e-2)."""
    from llm_quest_amd.qwen.qwen3_5.qwen3_5_vlm_model import Qwen3_5VLM
    from oracle.gen_golden import TINY_Q35_TEXT

    t = golden("qwen35_vlm_tiny")
    cfg = {**TINY_Q35_TEXT, **TINY_Q35_VISION, "llm_d_in": TINY_Q35_TEXT["emb_dim"], "image_token_id": 250, "dtype": BF16}
    sd = sub_dict(t, "sd.")
    vlm = Qwen3_5VLM(cfg)
    assert set(vlm.state_dict()) == set(sd)
    vlm.load_state_dict({k: (v.bool() if k.endswith("mask") else v) for k, v in sd.items()})
    vlm = vlm.cuda().train()
    ids, pix, am, gout = t["in.ids"].cuda(), t["in.pixels"].cuda(), t["in.attn_mask"].bool().cuda(), t["gout"].cuda()
    pid = vlm.compute_3d_position_ids(ids, vlm.get_feeds_3d_shape(pix), image_mask=ids == 250)
    assert torch.equal(pid.cpu(), t["pos3d"])
    logits = vlm(ids, image_pixels=pix, attn_mask=am)
    assert logits.dtype == BF16 and logits.shape == t["bf16.logits"].shape

    def within(mine, floor, what):
        tol = 1.5 * floor + (0.0 if floor >= 1e-2 else 2e-3)
        assert mine <= tol, f"{what}: mine {mine:.3e}, reference floor {floor:.3e}"

    within(rel_l2(logits, t["fp32.logits"]), rel_l2(t["bf16.logits"], t["fp32.logits"]), "logits vs the fp32 twin")
    (logits.float() * gout).sum().backward()
    checked = 0
    for name, p in vlm.named_parameters():
        if name.endswith("out_head.weight"):
            continue
        ref, twin = t["bf16.grad." + name], t["fp32.grad." + name]
        assert p.grad is not None and p.grad.shape == ref.shape, name
        if float(twin.float().abs().max()) == 0.0:  # unused rows / parameters: the gradient must be zero here too
            assert float(p.grad.float().abs().max()) == 0.0, name
            continue
        floor_g = rel_l2(ref, twin)
        # gate parameters with a handful of elements carry the bf16 rounding of softplus / sigmoid un-averaged (test_qwen35_text_gpu.py)
        few = ref.numel() <= 256 and any(s_ in name for s_ in ("log_A", "dt_bias", "w_alpha", "w_beta"))
        mine = rel_l2(p.grad, twin)
        assert mine <= (2.5 if few else 1.5) * floor_g + (0.0 if floor_g >= 1e-2 else 2e-3), f"{name}: mine {mine:.3e}, reference floor {floor_g:.3e}"
        checked += 1
    assert checked > 60
