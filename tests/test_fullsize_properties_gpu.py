"""Full-size checks (BASELINE config-4 shapes) through size-independent properties, where the CPU oracle would take
minutes: exact integer checksums, causality, normalisation, bit-exact gathers, run-to-run determinism."""

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32
B, S, NV, T, D, HQ, HKV, DH, FF, V = 8, 709, 197, 512, 1024, 16, 8, 128, 3072, 151_936


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from llm_quest_amd import kernels

    return kernels


def _ints(shape, lo, hi, seed):
    g = torch.Generator().manual_seed(seed)
    return torch.randint(lo, hi + 1, shape, generator=g).to(BF16)


@pytest.mark.parametrize("form", ["NT", "NN", "TN"])
def test_gemm_exact_integer_checksums(K, form):
    """Small-integer operands make every product and partial sum exact in fp32, so the GPU result must equal integer
    arithmetic BIT FOR BIT.  Checked at the gate-up projection's size via the checksum identities
        sum_n C[m,n] = A[m,:] . (sum_n B[n,:])     and     sum_m C[m,n] = (sum_m A[m,:]) . B[n,:]
    (O(MK+NK) on the host), plus a 64x64 corner computed directly."""
    from llm_quest_amd import _lib as L

    M, N, Kd = B * S, 2 * FF, D
    a = _ints((M, Kd), -3, 3, 1)
    b = _ints((N, Kd), -3, 3, 2)
    if form == "NT":
        c = K.gemm(L.GEMM_NT, a.cuda(), b.cuda(), out_dtype=F32)
    elif form == "NN":
        c = K.gemm(L.GEMM_NN, a.cuda(), b.t().contiguous().cuda(), out_dtype=F32)
    else:
        c = K.gemm(L.GEMM_TN, a.t().contiguous().cuda(), b.t().contiguous().cuda(), out_dtype=F32)
    c = c.cpu().to(torch.int64)
    ai, bi = a.to(torch.int64), b.to(torch.int64)
    assert torch.equal(c.sum(dim=1), ai @ bi.sum(dim=0))
    assert torch.equal(c.sum(dim=0), bi @ ai.sum(dim=0))
    assert torch.equal(c[:64, :64], ai[:64] @ bi[:64].t())
    assert torch.equal(c[-64:, -64:], ai[-64:] @ bi[-64:].t())


def test_lm_head_gemm_exact_at_vocab_size(K):
    from llm_quest_amd import _lib as L

    rows = 1024
    h = _ints((rows, D), -2, 2, 3)
    w = _ints((V, D), -2, 2, 4)
    logits = K.gemm(L.GEMM_NT, h.cuda(), w.cuda(), out_dtype=F32).cpu().to(torch.int64)
    hi, wi = h.to(torch.int64), w.to(torch.int64)
    assert torch.equal(logits.sum(dim=1), hi @ wi.sum(dim=0))
    assert torch.equal(logits[:, -100:], hi @ wi[-100:].t())


def test_attention_fullsize_invariants(K):
    g = torch.Generator().manual_seed(7)
    q = torch.randn(B * S, HQ * DH, generator=g).to(BF16).cuda()
    k = torch.randn(B * S, HKV * DH, generator=g).to(BF16).cuda()
    v = torch.randn(B * S, HKV * DH, generator=g).to(BF16).cuda()
    o, lse = K.attn_fwd(q, k, v, B, S, HQ, HKV, DH, causal=True)
    # determinism: the kernel has no atomics
    o2, _ = K.attn_fwd(q, k, v, B, S, HQ, HKV, DH, causal=True)
    assert torch.equal(o, o2)
    # causality: changing keys/values at positions >= 400 cannot change any output row < 400, bit for bit
    k3, v3 = k.clone().view(B, S, -1), v.clone().view(B, S, -1)
    k3[:, 400:] = torch.randn_like(k3[:, 400:])
    v3[:, 400:] = torch.randn_like(v3[:, 400:])
    o3, _ = K.attn_fwd(q, k3.view(B * S, -1), v3.view(B * S, -1), B, S, HQ, HKV, DH, causal=True)
    assert torch.equal(o.view(B, S, -1)[:, :400], o3.view(B, S, -1)[:, :400])
    assert not torch.equal(o.view(B, S, -1)[:, 400:], o3.view(B, S, -1)[:, 400:])
    # normalisation: with V == 1 every output element is sum(P)/sum(p) ~ 1 (P rounded to bf16 in the numerator only)
    ones = torch.ones_like(v)
    o1, _ = K.attn_fwd(q, k, ones, B, S, HQ, HKV, DH, causal=True)
    assert float((o1.float() - 1).abs().max()) < 2 ** -7
    # the first query attends to exactly one key: output == that value row, lse == its scaled score
    row0 = o.view(B, S, HQ, DH)[:, 0]
    vrow0 = v.view(B, S, HKV, DH)[:, 0].repeat_interleave(HQ // HKV, dim=1)
    assert torch.equal(row0, vrow0)
    s00 = (q.view(B, S, HQ, DH)[:, 0].float() * k.view(B, S, HKV, DH)[:, 0].repeat_interleave(2, dim=1).float()).sum(-1) * DH ** -0.5
    assert torch.allclose(lse[:, :, 0], s00, rtol=1e-5, atol=1e-4)
    # all-ones key mask == no mask, bit for bit
    km = torch.ones(B, S, dtype=torch.uint8, device="cuda")
    o4, _ = K.attn_fwd(q, k, v, B, S, HQ, HKV, DH, key_mask=km, causal=True)
    assert torch.equal(o, o4)


def test_attention_backward_fullsize_consistency(K):
    """dV of a non-causal all-to-one pattern and finite-difference-free identities: sum over queries of dO-weighted P."""
    g = torch.Generator().manual_seed(9)
    b_ = 2
    q = torch.randn(b_ * S, HQ * DH, generator=g).to(BF16).cuda()
    k = torch.randn(b_ * S, HKV * DH, generator=g).to(BF16).cuda()
    v = torch.randn(b_ * S, HKV * DH, generator=g).to(BF16).cuda()
    do = torch.randn(b_ * S, HQ * DH, generator=g).to(BF16).cuda()
    o, lse = K.attn_fwd(q, k, v, b_, S, HQ, HKV, DH, causal=True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, do, lse, b_, S, HQ, HKV, DH, dq, dk, dv, causal=True)
    dq2, dk2, dv2 = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, do, lse, b_, S, HQ, HKV, DH, dq2, dk2, dv2, causal=True)
    assert torch.equal(dq, dq2) and torch.equal(dk, dk2) and torch.equal(dv, dv2)  # deterministic (no atomics)
    # the last key is seen only by the last query: dV[last] = sum over the group's heads of P[last,last] * dO[last]
    p_last = torch.exp((q.view(b_, S, HQ, DH)[:, -1].float() * k.view(b_, S, HKV, DH)[:, -1].repeat_interleave(2, 1).float()).sum(-1) * DH ** -0.5 - lse[:, :, -1])
    expect = (p_last[..., None] * do.view(b_, S, HQ, DH)[:, -1].float()).view(b_, HKV, 2, DH).sum(2)
    got = dv.view(b_, S, HKV, DH)[:, -1].float()
    assert float((got - expect).norm() / expect.norm()) < 1e-2
    # softmax-Jacobian identity: every row of dS sums to zero  =>  sum_d dQ[q,d]*Q[q,d] == sum over keys of dK.K restricted..;
    # use the cheap global form: <dQ, Q> == <dK, K> (both equal sum_{q,k} dS[q,k] * S_raw[q,k])
    lhs = (dq.double() * q.double()).sum()
    rhs = (dk.double() * k.double()).sum()
    scale = (dq.double() * q.double()).abs().sum()
    assert abs(float(lhs - rhs)) < 1e-3 * float(scale)


def test_attention_backward_persistent_equals_per_block_workgroups(K):
    """With >= 512 (batch, head) pairs one workgroup walks ALL blocks of its pair (3-stage tile stream across block
    boundaries); below that there is one workgroup per block.  Same arithmetic in the same order -> bit-identical gradients.
    The small-batch form is the one the oracle parity tests exercise (tests/test_kernels_gpu.py)."""
    g = torch.Generator().manual_seed(21)
    b_ = 64  # 64 * 8 kv heads = 512 pairs -> persistent dK/dV;  64 * 16 = 1024 -> persistent dQ
    q = torch.randn(b_ * S, HQ * DH, generator=g).to(BF16).cuda()
    k = torch.randn(b_ * S, HKV * DH, generator=g).to(BF16).cuda()
    v = torch.randn(b_ * S, HKV * DH, generator=g).to(BF16).cuda()
    do = torch.randn(b_ * S, HQ * DH, generator=g).to(BF16).cuda()
    km = torch.ones(b_, S, dtype=torch.uint8)
    km[::3, S - 150 :] = 0  # ragged rows: padded keys at the end of every third sample
    km = km.cuda()
    o, lse = K.attn_fwd(q, k, v, b_, S, HQ, HKV, DH, key_mask=km, causal=True)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    K.attn_bwd(q, k, v, o, do, lse, b_, S, HQ, HKV, DH, dq, dk, dv, key_mask=km, causal=True)
    part = 16  # 16 * 16 = 256 < 512 pairs: one workgroup per block
    for i in range(0, b_, part):
        r = slice(i * S, (i + part) * S)
        dq2, dk2, dv2 = torch.empty_like(q[r]), torch.empty_like(k[r]), torch.empty_like(v[r])
        K.attn_bwd(q[r], k[r], v[r], o[r], do[r], lse[i : i + part], part, S, HQ, HKV, DH, dq2, dk2, dv2, key_mask=km[i : i + part], causal=True)
        assert torch.equal(dq[r], dq2) and torch.equal(dk[r], dk2) and torch.equal(dv[r], dv2), i
    assert torch.isfinite(dq.float()).all() and torch.isfinite(dk.float()).all() and torch.isfinite(dv.float()).all()


def test_fusion_embedding_patch_gathers_bit_exact_at_full_size(K):
    from llm_quest_amd.multimodal.vlm_engine import fuse_embeddings
    from oracle import index_ops

    g = torch.Generator().manual_seed(11)
    vis = torch.randn(B, NV, D, generator=g).to(BF16)
    txt = torch.randn(B, T, D, generator=g).to(BF16)
    assert torch.equal(fuse_embeddings(vis.cuda(), txt.cuda()).cpu(), torch.cat([vis, txt], dim=1))
    table = torch.randn(V, D, generator=g).to(BF16)
    ids = torch.randint(0, V, (B, T), generator=g)
    assert torch.equal(K.embedding_fwd(ids.cuda(), table.cuda()).cpu(), table[ids.reshape(-1)])
    img = torch.randn(B, 3, 224, 224, generator=g)
    rows = K.patchify(img.cuda(), 16, out_dtype=F32).cpu()
    idx = torch.from_numpy(index_ops.patch_gather_index(3, 224, 224, 16))
    for b_ in (0, B - 1):
        assert torch.equal(rows[b_ * 196 : (b_ + 1) * 196], img[b_].reshape(-1)[idx])


def test_full_size_step_loss_at_init_and_determinism():
    """Whole config-4 step at full model size (B=4): loss ~ ln(V) at random init, finite gradients, and two identical
    steps give bit-identical loss and gradients (the embedding scatter-add is the only atomic; it sums in fp32)."""
    import math

    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    import bench
    from llm_quest_amd.multimodal.vlm_engine import vlm_step_loss

    dev = torch.device("cuda", 0)
    vit, vit_cfg, ad, llm, llm_cfg = bench.build_models(dev)
    img, ids, mask = bench.synthetic_batch(4, dev, seed=5)

    def step():
        llm.zero_grad(set_to_none=True)
        ad.zero_grad(set_to_none=True)
        loss = vlm_step_loss(vit, llm, ad, img, ids, mask, hf_vit_model=False)
        loss.backward()
        return loss.detach().clone(), llm.trf_blocks[13].ffn.lin1.weight.grad.clone(), ad.adapter[0].weight.grad.clone()

    l1, g1, a1 = step()
    l2, g2, a2 = step()
    assert abs(float(l1) - math.log(V)) < 0.05
    assert torch.isfinite(g1.float()).all() and float(g1.float().norm()) > 0
    assert torch.equal(l1, l2) and torch.equal(g1, g2) and torch.equal(a1, a2)
