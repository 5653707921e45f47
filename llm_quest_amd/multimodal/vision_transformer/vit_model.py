"""ViT model on HIP kernels -- API of ``llm_quest/multimodal/vision_transformer/vit_model.py``."""

import torch
import torch.nn as nn

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd.multimodal.vision_transformer.vit_attention import bf16_cached, split3_cached
from llm_quest_amd.multimodal.vision_transformer.vit_transformer_block import LayerNorm, ViTTransformerBlock

BF16, F32 = torch.bfloat16, torch.float32


class PatchEmbedding2D(nn.Module):
    """Non-overlapping patches -> embeddings, CLS token prepended (reference: vit_model.py:19-89).

    ``conv_proj`` is kept as an ``nn.Conv2d`` so the state_dict keys/shapes are the reference's, but it is never
    called: the patches are gathered by a coalesced im2row kernel (bit-exact index map, K ordered (c, i, j)) and
    projected by the MFMA GEMM against the (emb, C*P*P) view of the conv weight.
    """

    def __init__(self, img_width, img_height, patch_size, num_channels, emb_dim):
        super().__init__()
        assert img_width % patch_size == 0, f"Image width {img_width} not divisible by patch size {patch_size}"
        assert img_height % patch_size == 0, f"Image height {img_height} not divisible by patch size {patch_size}"
        self.img_width, self.img_height, self.patch_size = img_width, img_height, patch_size
        self.num_patches = (img_width * img_height) // patch_size**2
        self.conv_proj = nn.Conv2d(num_channels, emb_dim, kernel_size=(patch_size, patch_size), stride=(patch_size, patch_size), padding=0, bias=True)
        self.cls_token = nn.Parameter(torch.randn(1, 1, emb_dim))

    def project(self, x, f32=False):
        """(b, c, h, w) fp32 -> patch projections fp32 [b*num_patches, emb] (bias added, no CLS).  ``f32``: pixels and weights as split bf16
        operands (fp32-grade product, the frozen tower at the reference's precision)."""
        assert x.shape[2] == self.img_width and x.shape[3] == self.img_height, (
            f"Input image shape {x.shape} does not match expected shape {self.img_width}x{self.img_height}"
        )
        if f32:
            rows = K.split3(K.patchify(x.contiguous().to(F32), self.patch_size, out_dtype=F32))
            return K.gemm(L.GEMM_NT, rows, split3_cached(self, "wconv3", [self.conv_proj.weight]), bias=self.conv_proj.bias.detach(), out_dtype=F32)
        rows = K.patchify(x.contiguous().to(F32), self.patch_size, out_dtype=BF16)
        w = bf16_cached(self, "wconv", [self.conv_proj.weight])
        return K.gemm(L.GEMM_NT, rows, w, bias=self.conv_proj.bias.detach(), out_dtype=F32)

    def forward(self, x):
        """(b, c, h, w) -> (b, num_patches + 1, emb) fp32, CLS row first (reference :68-89).  Autograd node: the backward writes the
        projection's weight / bias gradients and the CLS token's; pixels receive no gradient."""
        from llm_quest_amd.multimodal.vision_transformer import vit_train as T

        b, emb = x.shape[0], self.cls_token.shape[-1]

        def fwd(px):
            assert px.shape[2] == self.img_width and px.shape[3] == self.img_height, (
                f"Input image shape {px.shape} does not match expected shape {self.img_width}x{self.img_height}"
            )
            rows = K.patchify(px.contiguous().to(F32), self.patch_size, out_dtype=BF16)
            w = bf16_cached(self, "wconv", [self.conv_proj.weight])
            proj = K.gemm(L.GEMM_NT, rows, w, bias=self.conv_proj.bias.detach(), out_dtype=F32)
            out = torch.empty((b, self.num_patches + 1, emb), dtype=F32, device=px.device)
            flat = out.view(b, -1)
            K.copy2d(self.cls_token.detach().reshape(1, emb).to(F32).expand(b, emb).contiguous(), flat[:, :emb])
            K.copy2d(proj.view(b, -1), flat[:, emb:])
            return out, (rows,)

        def bwd(saved, dy):
            g = T.as_f32_rows(dy).view(b, -1)
            T._acc(self.cls_token, K.colsum(g[:, :emb]))
            dproj = torch.empty((b, self.num_patches * emb), dtype=F32, device=g.device)
            K.copy2d(g[:, emb:], dproj)
            dproj_b = K.cast(dproj, BF16).view(b * self.num_patches, emb)
            T._wgrad(self.conv_proj.weight, dproj_b, saved[0])
            T._bgrad(self.conv_proj.bias, dproj_b)
            return None

        return T.run_piece(self, x, fwd, bwd)


TOWER_PRECISIONS = ("bf16", "fp32")


def tower_precision(vit):
    """Arithmetic of the FROZEN tower's hidden states (``ViTModel(images, output_hidden_states=True)`` without grad: the vision half of the early-fusion
    step).  "fp32": what the reference computes there (multimodal/vlm_engine.py:99-104 calls the ViT outside autocast) -- split-bf16 GEMMs + exact-fp32
    attention, hidden states within 1e-4 of the fp32 reference; "bf16": bf16 MFMA operands with an fp32 residual stream (the dtype flow of the
    reference's ViT TRAINING path under autocast), hidden states within ~1e-2.  Per model (``vit.tower_precision = ...``) or
    ``MI355_VIT_TOWER``; DESIGN.md section 4 has the measured price of each."""
    import os

    val = getattr(vit, "tower_precision", None) or os.environ.get("MI355_VIT_TOWER", DEFAULT_TOWER_PRECISION)
    if val not in TOWER_PRECISIONS:
        raise ValueError(f"tower precision must be one of {TOWER_PRECISIONS}, got {val!r}")
    return val


DEFAULT_TOWER_PRECISION = "bf16"  # measured (profiles/r05_notes.md): fp32 costs the early-fusion step 4.0 % (447.1 -> 465.2 ms at per-GPU batch 160)


class ViTModel(nn.Module):
    """Patch embed + learned positions + pre-LN encoder + final LN (+ class head) (reference: vit_model.py:92-160)."""

    def __init__(self, cfg):
        super().__init__()
        self.patch_embedding = PatchEmbedding2D(cfg["img_width"], cfg["img_height"], cfg["patch_size"], cfg["num_channels"], cfg["emb_dim"])
        self.pos_embedding = nn.Parameter(torch.randn(1, self.patch_embedding.num_patches + 1, cfg["emb_dim"]))
        self.dropout = nn.Dropout(cfg["drop_rate"])
        self.transformer_blocks = nn.ModuleList([ViTTransformerBlock(cfg) for _ in range(cfg["n_layers"])])
        self.final_ln = LayerNorm(cfg["emb_dim"])
        self.classifier = nn.Linear(cfg["emb_dim"], cfg["num_classes"])

    def forward(self, x, output_hidden_states=False):
        L.require_gpu(x)
        from llm_quest_amd.multimodal.vision_transformer import vit_train

        if vit_train.needs_training_path(self):
            # trainable ViT (BASELINE config 2): one autograd node, bf16 logits as under the reference's autocast
            return vit_train.run_train(self, x, output_hidden_states)
        b = x.shape[0]
        pe = self.patch_embedding
        s, d = pe.num_patches + 1, self.pos_embedding.shape[-1]
        f32 = output_hidden_states and tower_precision(self) == "fp32"
        proj = pe.project(x, f32)
        # CLS row + positional embedding in one pass (vit_model.py:86-87,145)
        h = K.vit_embed_assemble(proj, pe.cls_token.detach().reshape(-1).contiguous(), self.pos_embedding.detach().reshape(s, d).contiguous(), b, s, d)
        h2 = h.view(b * s, d)
        for blk in self.transformer_blocks:
            h2 = blk.run_f32(h2, b, s) if f32 else blk.run(h2, b, s)
        if output_hidden_states:
            return self.final_ln.normalize(h2, F32).view(b, s, d)
        cls_rows = h2.view(b, s, d)[:, 0].contiguous()
        cls_n = self.final_ln.normalize(cls_rows, BF16)
        wc = bf16_cached(self, "wcls", [self.classifier.weight])
        return K.gemm(L.GEMM_NT, cls_n, wc, bias=self.classifier.bias.detach(), out_dtype=F32)
