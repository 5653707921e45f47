"""Input pipeline on the step's left edge -- batch contract of ``MultimodalDataset``
(llm_quest/dataset.py:295-383) with the per-sample arithmetic on HIP kernels.

Upstream does, per sample on the host: PIL resize to (image_size, image_size) (bilinear), ``ToTensor``, ImageNet ``Normalize``,
tokenise ``caption + eos``, truncate / pad to ``max_caption_len`` with pad = eos, bool mask.  Here the host only decodes the
image to a uint8 HWC array and tokenises the caption; everything else runs on the GPU:

  * ``MultimodalDataset[i]`` keeps the reference's return contract ({"image" (3, s, s) fp32, "input_ids" (L,) int64,
    "attention_mask" (L,) bool}) with device tensors;
  * ``MultimodalDataset.batches(batch_size)`` is the fast path for ``vlm_training_loop_simple``: raw bytes are staged in pinned
    host memory and copied on a side stream one batch AHEAD of the consumer, the resize / normalise / pad kernels run on that
    stream too, and an event hands the finished batch to the compute stream -- the step never waits for PCIe.

Resize is Pillow's fixed-point resampler reproduced bit for bit (``resize_tables`` = its coefficient tables, built once per
source size in float64 exactly as ``precompute_coeffs`` does).  No CPU fallback: without the HIP library / a GPU this raises.
"""

import numpy as np
import torch

from . import _lib as L

PRECISION_BITS = 22
IMAGENET_MEAN = (0.485, 0.456, 0.406)
IMAGENET_STD = (0.229, 0.224, 0.225)

_tables = {}


def resize_tables(in_size, out_size):
    """(bounds int32 [out, 2], kk int32 [out, ksize]) of Pillow's bilinear filter for one axis (Resample.c: precompute_coeffs +
    normalize_coeffs_8bpc), as host numpy arrays; vectorised, float64, same operation order as the C code."""
    key = (int(in_size), int(out_size))
    t = _tables.get(key)
    if t is not None:
        return t
    scale = float(np.float32(in_size)) / out_size
    filterscale = max(scale, 1.0)
    support = filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    xx = np.arange(out_size, dtype=np.float64)
    center = (xx + 0.5) * scale
    xmin = np.maximum(np.trunc(center - support + 0.5).astype(np.int64), 0)
    xmax = np.minimum(np.trunc(center + support + 0.5).astype(np.int64), in_size) - xmin
    x = np.arange(ksize, dtype=np.float64)[None, :]
    a = np.abs((x + xmin[:, None] - center[:, None] + 0.5) * (1.0 / filterscale))
    w = np.where(a < 1.0, 1.0 - a, 0.0)
    w = np.where(x < xmax[:, None], w, 0.0)
    ww = np.zeros(out_size, dtype=np.float64)
    for j in range(ksize):  # sequential accumulation, as the C loop (the order of a float sum matters)
        ww = ww + w[:, j]
    w = np.where(ww[:, None] != 0.0, w / np.where(ww[:, None] != 0.0, ww[:, None], 1.0), w)
    kk = np.trunc(0.5 + w * (1 << PRECISION_BITS)).astype(np.int32)  # weights are non-negative for the bilinear filter
    bounds = np.stack([xmin, xmax], axis=1).astype(np.int32)
    t = (np.ascontiguousarray(bounds), np.ascontiguousarray(kk))
    _tables[key] = t
    return t


class _DeviceTables:
    """Coefficient tables resident on the device, per (in_size, out_size)."""

    def __init__(self, device):
        self.device = device
        self.cache = {}

    def get(self, in_size, out_size):
        key = (in_size, out_size)
        t = self.cache.get(key)
        if t is None:
            b, k = resize_tables(in_size, out_size)
            t = (torch.from_numpy(b).to(self.device), torch.from_numpy(k).to(self.device), k.shape[1])
            self.cache[key] = t
        return t


def image_transform_into(img_u8_dev, out_chw, tables, image_size, mean_dev, std_dev, scratch=None):
    """img_u8_dev: uint8 (H, W, 3) on the device -> out_chw fp32 (3, s, s) (resize + ToTensor [+ Normalize]); all on the current stream."""
    L.require_gpu(img_u8_dev, out_chw)
    if img_u8_dev.dtype != torch.uint8 or img_u8_dev.dim() != 3 or img_u8_dev.shape[2] != 3 or not img_u8_dev.is_contiguous():
        raise ValueError(f"image must be a contiguous uint8 (H, W, 3) RGB array, got {img_u8_dev.dtype} {tuple(img_u8_dev.shape)}")
    if out_chw.dtype != torch.float32 or tuple(out_chw.shape) != (3, image_size, image_size) or not out_chw.is_contiguous():
        raise ValueError("output must be contiguous fp32 (3, s, s)")
    H, W, C = img_u8_dev.shape
    bh, kh, ksh = tables.get(W, image_size)
    bv, kv, ksv = tables.get(H, image_size)
    tmp = scratch if scratch is not None and scratch.numel() >= H * image_size * C else torch.empty(H * image_size * C, dtype=torch.uint8, device=img_u8_dev.device)
    L.call("mi355_resize_h_u8", H, W, image_size, C, L.ptr(img_u8_dev), W * C, L.ptr(bh), L.ptr(kh), ksh, L.ptr(tmp))
    L.call("mi355_resize_v_normalize", H, image_size, image_size, C, L.ptr(tmp), L.ptr(bv), L.ptr(kv), ksv, L.ptr(mean_dev), L.ptr(std_dev), L.ptr(out_chw))
    return out_chw


def pad_tokens(flat_ids_dev, offsets_dev, B, max_len, pad_id):
    """-> (input_ids int64 [B, L], attention_mask bool [B, L])."""
    L.require_gpu(flat_ids_dev, offsets_dev)
    if flat_ids_dev.dtype != torch.int64 or offsets_dev.dtype != torch.int64 or offsets_dev.numel() != B + 1:
        raise ValueError("pad_tokens: flat ids int64, offsets int64 [B+1]")
    ids = torch.empty((B, max_len), dtype=torch.int64, device=flat_ids_dev.device)
    mask = torch.empty((B, max_len), dtype=torch.uint8, device=flat_ids_dev.device)
    L.call("mi355_pad_tokens", B, max_len, L.ptr(flat_ids_dev), L.ptr(offsets_dev), int(pad_id), L.ptr(ids), L.ptr(mask))
    return ids, mask.view(torch.bool)


def _as_rgb_u8(image):
    """PIL image / numpy array -> contiguous uint8 (H, W, 3) host array (decode only; `image.convert("RGB")` as upstream :361-362)."""
    if hasattr(image, "mode"):  # PIL
        if image.mode != "RGB":
            image = image.convert("RGB")
        image = np.asarray(image)
    a = np.ascontiguousarray(image)
    if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3:
        raise ValueError(f"expected an RGB uint8 (H, W, 3) image, got {a.dtype} {a.shape}")
    return a


class MultimodalDataset(torch.utils.data.Dataset):
    """Image-caption pairs (reference: dataset.py:295-383; same constructor, same item dict), GPU-side transforms.

    ``tokenizer``: anything with ``eos_token``, ``eos_token_id`` and ``tokenizer(text)["input_ids"]`` (Hugging Face tokenizers do).
    """

    def __init__(self, hf_dataset_split, tokenizer, image_size=224, max_caption_len=128, image_key="image", caption_key="caption_0",
                 standardize=True, device="cuda"):
        self.tokenizer = tokenizer
        self.tokenizer.pad_token = self.tokenizer.eos_token
        self.dataset = hf_dataset_split
        self.image_size = image_size
        self.max_caption_len = max_caption_len
        self.image_key = image_key
        self.caption_key = caption_key
        self.standardize = standardize
        self.device = torch.device(device)
        self._dev_state = None

    def __len__(self):
        return len(self.dataset)

    # ------------------------------------------------------------------ host side: decode + tokenise only
    def raw_item(self, idx):
        item = self.dataset[idx]
        ids = self.tokenizer(item[self.caption_key] + self.tokenizer.eos_token)["input_ids"]
        return _as_rgb_u8(item[self.image_key]), [int(t) for t in ids]

    def _state(self):
        if self._dev_state is None:
            if not torch.cuda.is_available():
                raise RuntimeError("MultimodalDataset transforms run on an MI355X (HIP) device; there is no CPU fallback for this path")
            L.load()
            mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=self.device) if self.standardize else None
            std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=self.device) if self.standardize else None
            self._dev_state = (_DeviceTables(self.device), mean, std)
        return self._dev_state

    # ------------------------------------------------------------------ reference contract: one item
    def __getitem__(self, idx):
        img, ids = self.raw_item(idx)
        tables, mean, std = self._state()
        out = torch.empty((3, self.image_size, self.image_size), dtype=torch.float32, device=self.device)
        image_transform_into(torch.from_numpy(img).to(self.device), out, tables, self.image_size, mean, std)
        flat = torch.tensor(ids if ids else [0], dtype=torch.int64, device=self.device)
        offs = torch.tensor([0, len(ids)], dtype=torch.int64, device=self.device)
        tok, mask = pad_tokens(flat, offs, 1, self.max_caption_len, self.tokenizer.eos_token_id)
        return {"image": out, "input_ids": tok[0], "attention_mask": mask[0]}

    # ------------------------------------------------------------------ fast path: prefetched GPU batches
    def batches(self, batch_size, indices=None, drop_last=False):
        """Yields {"image" (B, 3, s, s) fp32, "input_ids" (B, L) int64, "attention_mask" (B, L) bool} on the device, one batch
        prefetched: while the consumer runs step i, batch i+1 is staged in pinned memory, copied and transformed on a side stream."""
        tables, mean, std = self._state()
        order = list(range(len(self))) if indices is None else list(indices)
        chunks = [order[i : i + batch_size] for i in range(0, len(order), batch_size)]
        if drop_last and chunks and len(chunks[-1]) < batch_size:
            chunks.pop()
        side = torch.cuda.Stream(device=self.device)

        def produce(chunk):
            raws = [self.raw_item(i) for i in chunk]
            B = len(raws)
            nbytes = sum(r[0].size for r in raws)
            stage = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
            pos, views = 0, []
            for img, _ in raws:
                stage[pos : pos + img.size] = torch.from_numpy(img.reshape(-1))
                views.append((pos, img.shape))
                pos += img.size
            flat_ids = [t for _, ids in raws for t in ids] or [0]
            offs = np.cumsum([0] + [len(ids) for _, ids in raws])
            ids_host = torch.tensor(flat_ids, dtype=torch.int64).pin_memory()
            offs_host = torch.from_numpy(offs.astype(np.int64)).pin_memory()
            with torch.cuda.stream(side):
                dev_bytes = stage.to(self.device, non_blocking=True)
                ids_dev = ids_host.to(self.device, non_blocking=True)
                offs_dev = offs_host.to(self.device, non_blocking=True)
                images = torch.empty((B, 3, self.image_size, self.image_size), dtype=torch.float32, device=self.device)
                scratch = torch.empty(max(v[1][0] for v in views) * self.image_size * 3, dtype=torch.uint8, device=self.device)
                for b, (p, shp) in enumerate(views):
                    n = shp[0] * shp[1] * shp[2]
                    image_transform_into(dev_bytes[p : p + n].view(shp), images[b], tables, self.image_size, mean, std, scratch)
                tok, mask = pad_tokens(ids_dev, offs_dev, B, self.max_caption_len, self.tokenizer.eos_token_id)
                done = torch.cuda.Event()
                done.record(side)
            return {"image": images, "input_ids": tok, "attention_mask": mask}, done, (stage, ids_host, offs_host, dev_bytes, scratch)

        pending = produce(chunks[0]) if chunks else None
        for i in range(len(chunks)):
            batch, done, keep = pending
            pending = produce(chunks[i + 1]) if i + 1 < len(chunks) else None
            torch.cuda.current_stream(self.device).wait_event(done)
            for t in batch.values():
                t.record_stream(torch.cuda.current_stream(self.device))
            yield batch
            del keep
