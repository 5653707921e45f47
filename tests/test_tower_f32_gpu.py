"""The frozen vision tower at the reference's fp32 precision (csrc/tower_f32.hip; reference multimodal/vlm_engine.py:99-104 calls the ViT outside
autocast): split-bf16 operands through the bf16 MFMA GEMMs, exact-fp32 attention, against fp64 arithmetic and against the reference fixture."""

import os

import pytest
import torch

pytestmark = pytest.mark.gpu
BF16, F32, F64 = torch.bfloat16, torch.float32, torch.float64
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def K():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from llm_quest_amd import kernels

    return kernels


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


@pytest.mark.parametrize("rows,width,pitch", [(5, 64, 64), (197, 768, 768), (33, 3072, 3072), (64, 768, 2304)])
def test_split3_is_the_two_term_bf16_expansion_bit_for_bit(K, rows, width, pitch):
    g = torch.Generator().manual_seed(rows + width)
    full = torch.randn(rows, pitch, generator=g) * torch.logspace(-3, 3, pitch)[None, :]
    full[0, :8] = torch.tensor([0.0, -0.0, 1.0, 2.0 ** -130, 3.0e38, -3.0e38, 1.0 + 2.0 ** -9, 1.0 - 2.0 ** -10])  # zeros, a denormal, huge, ties
    x = full[:, :width]
    hi = x.to(BF16)
    lo = (x - hi.float()).to(BF16)
    xa = full.cuda()[:, :width]  # a row-strided view when pitch > width
    act = K.split3(xa).cpu()
    wgt = K.split3(xa, weight_order=True).cpu()
    assert torch.equal(act.view(torch.int16), torch.cat([hi, lo, hi], dim=1).view(torch.int16))
    assert torch.equal(wgt.view(torch.int16), torch.cat([hi, hi, lo], dim=1).view(torch.int16))
    # the expansion carries 16 significant bits
    ok = x.abs() < 1e37
    assert float(((hi.double() + lo.double() - x.double()).abs() / x.double().abs().clamp_min(1e-30))[ok].max()) < 2.0 ** -15


@pytest.mark.parametrize("M,N,Kd,gelu", [(197 * 3, 2304, 768, False), (197 * 3, 3072, 768, True), (197 * 3, 768, 3072, False), (37, 64, 64, True)])
def test_split_operand_gemm_is_fp32_grade(K, M, N, Kd, gelu):
    """One NT GEMM over K' = 3K of [hi | lo | hi] x [hi | hi | lo] against fp64 (bias, GELU and residual epilogues as the tower uses them): 1e-5,
    where plain bf16 operands give 3e-3."""
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(M + N)
    a = torch.randn(M, Kd, generator=g)
    w = torch.randn(N, Kd, generator=g) * Kd ** -0.5
    bias = torch.randn(N, generator=g)
    res = torch.randn(M, N, generator=g)
    ref = a.double() @ w.double().t() + bias.double()
    if gelu:
        ref = torch.nn.functional.gelu(ref)
    else:
        ref = ref + res.double()
    out = K.gemm(L.GEMM_NT, K.split3(a.cuda()), K.split3(w.cuda(), weight_order=True), bias=bias.cuda(), gelu=gelu, residual=None if gelu else res.cuda(), out_dtype=F32)
    plain = K.gemm(L.GEMM_NT, a.cuda().to(BF16), w.cuda().to(BF16), bias=bias.cuda(), gelu=gelu, residual=None if gelu else res.cuda(), out_dtype=F32)
    assert _rel(out.cpu(), ref) < 1e-5, _rel(out.cpu(), ref)
    assert _rel(plain.cpu(), ref) > 20 * _rel(out.cpu(), ref)


@pytest.mark.parametrize("rows,width", [(197 * 3 + 1, 768), (130, 64), (65, 3072)])
def test_producers_write_the_split_operand_themselves_bit_for_bit(K, rows, width):
    """Round 6: LayerNorm and the Linear (+ GELU) epilogue leave their fp32 result as [hi | lo | hi] directly (MI355_DT_SPLIT3): the same bits as the fp32 output
    followed by mi355_split3_bf16, without the fp32 round trip."""
    from llm_quest_amd import _lib as L

    g = torch.Generator().manual_seed(rows + width)
    x = (torch.randn(rows, width, generator=g) * 3 + 0.5).cuda()
    sc, sh = (1 + 0.1 * torch.randn(width, generator=g)).cuda(), (0.1 * torch.randn(width, generator=g)).cuda()
    for mode in (0, 1):
        want = K.split3(K.layernorm_fwd(x, sc, sh, out_dtype=F32, mode=mode))
        got = K.layernorm_fwd(x, sc, sh, out_dtype="split3", mode=mode)
        assert got.shape == (rows, 3 * width) and torch.equal(got.view(torch.int16), want.view(torch.int16))
    N = 136 if width == 64 else 256
    a = K.split3(x)
    w = K.split3((torch.randn(N, width, generator=g) / width**0.5).cuda(), weight_order=True)
    bias = torch.randn(N, generator=g).cuda()
    for gelu in (False, True):
        want = K.split3(K.gemm(L.GEMM_NT, a, w, bias=bias, gelu=gelu, out_dtype=F32))
        got = K.gemm(L.GEMM_NT, a, w, bias=bias, gelu=gelu, split3_out=True)
        assert got.shape == (rows, 3 * N) and torch.equal(got.view(torch.int16), want.view(torch.int16)), (rows, width, gelu)
    with pytest.raises(ValueError, match="split3_out"):
        K.gemm(L.GEMM_NT, a, w, residual=torch.zeros(rows, N, device="cuda"), split3_out=True)


@pytest.mark.parametrize("B,S,H", [(2, 5, 2), (3, 197, 12), (1, 256, 3), (2, 33, 1), (1, 128, 2), (1, 129, 2), (1, 257, 2), (1, 288, 1)])
def test_attn_f32_against_fp64(K, B, S, H):
    """softmax(q k^T / sqrt(d)) v, every key visible (vit_attention.py:73-82), on row-strided views of a fused qkv matrix."""
    D = 64
    g = torch.Generator().manual_seed(S * H)
    qkv = torch.randn(B * S, 3 * H * D, generator=g) * 1.5
    q, k, v = (qkv[:, i * H * D : (i + 1) * H * D] for i in range(3))
    hd = lambda t: t.double().view(B, S, H, D).transpose(1, 2)
    p = torch.softmax(hd(q) @ hd(k).transpose(-1, -2) * D ** -0.5, dim=-1)
    ref = (p @ hd(v)).transpose(1, 2).reshape(B * S, H * D)
    dq = qkv.cuda()
    out = K.attn_f32_fwd(dq[:, : H * D], dq[:, H * D : 2 * H * D], dq[:, 2 * H * D :], B, S, H, D)
    assert _rel(out.cpu(), ref) < 2e-6, _rel(out.cpu(), ref)
    assert float((out.cpu().double() - ref).abs().max()) < 1e-5
    out2 = K.attn_f32_fwd(dq[:, : H * D], dq[:, H * D : 2 * H * D], dq[:, 2 * H * D :], B, S, H, D)
    assert torch.equal(out, out2)


def test_attn_f32_rejects_what_it_does_not_cover(K):
    x = torch.zeros(289, 64, dtype=F32, device="cuda")
    with pytest.raises(RuntimeError, match="288 keys"):
        K.attn_f32_fwd(x, x, x, 1, 289, 1, 64)
    y = torch.zeros(8, 128, dtype=F32, device="cuda")
    with pytest.raises(RuntimeError, match="head_dim"):
        K.attn_f32_fwd(y, y, y, 1, 8, 1, 128)


def test_frozen_tower_fp32_reproduces_the_reference_fixture():
    """Tiny ViT (the fixture the reference itself produced in fp32, oracle/gen_golden.py::gen_vit): hidden states of the fp32-grade tower within 2e-5,
    the bf16-operand tower at its 1e-2; the switch is per model or MI355_VIT_TOWER."""
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")
    from safetensors.torch import load_file

    from llm_quest_amd.multimodal.vision_transformer import vit_model as VM
    from oracle.gen_golden import TINY_VIT

    t = load_file(os.path.join(GOLD, "vit_tiny.safetensors"))
    sd = {k[len("sd."):]: v for k, v in t.items() if k.startswith("sd.")}
    vit = VM.ViTModel(dict(TINY_VIT))
    vit.load_state_dict(sd)
    vit = vit.cuda().eval()
    img = t["in.image"].cuda()
    want = t["out.hidden"]
    with torch.no_grad():
        vit.tower_precision = "fp32"
        h32 = vit(img, output_hidden_states=True).cpu()
        vit.tower_precision = "bf16"
        h16 = vit(img, output_hidden_states=True).cpu()
    assert h32.dtype == F32 and h32.shape == want.shape
    assert _rel(h32, want) < 2e-5, _rel(h32, want)
    assert 1e-4 < _rel(h16, want) < 2e-2, _rel(h16, want)
    vit.tower_precision = "fp16"
    with pytest.raises(ValueError, match="tower precision"), torch.no_grad():
        vit(img, output_hidden_states=True)
