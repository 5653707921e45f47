"""CPU restatement of the input pipeline on the step's left edge: the batch contract of
``MultimodalDataset`` (llm_quest/dataset.py:295-383).  TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

The arithmetic of that path lives in third-party code that is NOT under /root/reference:
  * ``transforms.Resize((s, s))`` on a PIL image = ``PIL.Image.resize(..., BILINEAR)``: Pillow's two-pass fixed-point resampler
    (src/libImaging/Resample.c: precompute_coeffs, normalize_coeffs_8bpc, ImagingResampleHorizontal_8bpc / Vertical_8bpc).
    Pillow 12.2.0 is importable in this image, so the restatement below is PINNED against Pillow itself
    (tests/test_pipeline_cpu.py) and against committed vectors (tests/golden/pipeline.safetensors, oracle/gen_golden.py::gen_pipeline).
  * ``transforms.ToTensor`` / ``transforms.Normalize`` (torchvision, absent here; pinned upstream only as ``torchvision`` in
    pyproject.toml): published semantics uint8 HWC -> float32 CHW ``/ 255`` then ``(x - mean) / std`` per channel in fp32.
  * the tokenizer call ``tokenizer(caption + eos, truncation=True, max_length=L, padding="max_length")`` with pad = eos
    (dataset.py:337, 367-373): keep the first L ids, pad on the right with the eos id, mask = 1 on kept ids.
"""

import numpy as np
import torch

PRECISION_BITS = 32 - 8 - 2  # Resample.c
IMAGENET_MEAN = (0.485, 0.456, 0.406)  # dataset.py:349
IMAGENET_STD = (0.229, 0.224, 0.225)


def bilinear_coeffs(in_size, out_size):
    """precompute_coeffs + normalize_coeffs_8bpc for the bilinear filter (support 1.0) over the full axis [0, in_size).
    Returns (bounds int32 [out, 2] = (first input index, count), kk int32 [out, ksize] fixed-point weights)."""
    scale = float(np.float32(in_size) - np.float32(0.0)) / out_size
    filterscale = max(scale, 1.0)
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)  # C cast: truncation toward zero
        xmin = max(xmin, 0)
        xmax = min(int(center + support + 0.5), in_size) - xmin
        w = np.zeros(ksize, dtype=np.float64)
        ww = 0.0
        for x in range(xmax):
            a = abs((x + xmin - center + 0.5) * ss)
            w[x] = 1.0 - a if a < 1.0 else 0.0
            ww += w[x]
        if ww != 0.0:
            w[:xmax] /= ww
        for x in range(ksize):
            v = w[x] * (1 << PRECISION_BITS)
            kk[xx, x] = int(-0.5 + v) if w[x] < 0 else int(0.5 + v)
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(acc):
    return np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, out_h, out_w):
    """img uint8 (H, W, C) -> uint8 (out_h, out_w, C): horizontal pass, then vertical pass, each rounding to uint8 (ImagingResample);
    a pass whose size does not change is skipped, as Pillow does."""
    h, w, _ = img.shape
    cur = img
    if w != out_w:
        bounds, kk = bilinear_coeffs(w, out_w)
        out = np.empty((h, out_w, img.shape[2]), dtype=np.uint8)
        for xx in range(out_w):
            x0, n = bounds[xx]
            acc = np.full((h, img.shape[2]), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for x in range(n):
                acc += cur[:, x0 + x, :].astype(np.int64) * int(kk[xx, x])
            out[:, xx, :] = _clip8(acc)
        cur = out
    if h != out_h:
        bounds, kk = bilinear_coeffs(h, out_h)
        out = np.empty((out_h, cur.shape[1], img.shape[2]), dtype=np.uint8)
        for yy in range(out_h):
            y0, n = bounds[yy]
            acc = np.full((cur.shape[1], img.shape[2]), 1 << (PRECISION_BITS - 1), dtype=np.int64)
            for y in range(n):
                acc += cur[y0 + y].astype(np.int64) * int(kk[yy, y])
            out[yy] = _clip8(acc)
        cur = out
    return cur


def to_tensor_normalize(img_u8, standardize=True):
    """ToTensor (+ Normalize with the ImageNet statistics hard-coded upstream): uint8 HWC -> float32 CHW."""
    x = torch.from_numpy(np.ascontiguousarray(img_u8)).permute(2, 0, 1).contiguous().to(torch.float32).div(255)
    if standardize:
        mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32).view(-1, 1, 1)
        std = torch.tensor(IMAGENET_STD, dtype=torch.float32).view(-1, 1, 1)
        x = x.sub(mean).div(std)
    return x


def image_transform(img_u8, image_size=224, standardize=True):
    """MultimodalDataset.transform on an RGB uint8 (H, W, 3) array (dataset.py:341-358)."""
    return to_tensor_normalize(resize_bilinear_u8(img_u8, image_size, image_size), standardize)


def pad_caption(ids, max_len, pad_id):
    """ids: the token ids of ``caption + eos``.  Returns (input_ids int64 [max_len], attention_mask bool [max_len])."""
    kept = list(ids)[:max_len]
    out = torch.full((max_len,), pad_id, dtype=torch.int64)
    out[: len(kept)] = torch.tensor(kept, dtype=torch.int64)
    mask = torch.zeros(max_len, dtype=torch.bool)
    mask[: len(kept)] = True
    return out, mask
