"""Input pipeline, CPU side: the oracle's restatement of Pillow's resampler against committed Pillow
vectors and against Pillow itself, the host-built coefficient tables against the oracle's, and the pad / mask contract."""

import numpy as np
import pytest
import torch

from oracle import pipeline as P


def test_oracle_resize_matches_pillow_vectors(golden):
    t = golden("pipeline")
    for name in ("down", "up", "one_pass", "to224"):
        src, want = t[f"resize.{name}.in"].numpy(), t[f"resize.{name}.out"].numpy()
        got = P.resize_bilinear_u8(src, want.shape[0], want.shape[1])
        assert np.array_equal(got, want), name  # bit-exact


def test_oracle_resize_matches_live_pillow():
    Image = pytest.importorskip("PIL.Image")
    rng = np.random.default_rng(7)
    for h, w in ((333, 500), (224, 224), (50, 61), (500, 375), (224, 300), (97, 224), (1, 1), (225, 223)):
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ref = np.array(Image.fromarray(img).resize((224, 224), Image.BILINEAR))
        assert np.array_equal(P.resize_bilinear_u8(img, 224, 224), ref), (h, w)


def test_host_tables_equal_oracle_tables():
    from llm_quest_amd.dataset import resize_tables

    for n in (500, 333, 224, 100, 80, 97, 1, 3, 640, 225, 223, 1024):
        for out in (224, 64):
            b1, k1 = resize_tables(n, out)
            b2, k2 = P.bilinear_coeffs(n, out)
            assert np.array_equal(b1, b2) and np.array_equal(k1, k2), (n, out)
            assert b1.dtype == np.int32 and k1.dtype == np.int32


def test_to_tensor_normalize_and_pad_contract():
    img = np.arange(2 * 2 * 3, dtype=np.uint8).reshape(2, 2, 3) * 20
    x = P.to_tensor_normalize(img, standardize=False)
    assert x.shape == (3, 2, 2) and x.dtype == torch.float32 and float(x[1, 0, 1]) == np.float32(80) / np.float32(255)
    y = P.to_tensor_normalize(img)
    assert torch.equal(y[0], (x[0] - torch.tensor(0.485)) / torch.tensor(0.229))
    ids, mask = P.pad_caption([5, 6, 7], 5, 99)
    assert ids.tolist() == [5, 6, 7, 99, 99] and mask.tolist() == [True, True, True, False, False]
    ids, mask = P.pad_caption(list(range(9)), 4, 99)  # truncation drops the tail (the eos with it), as the tokenizer call does
    assert ids.tolist() == [0, 1, 2, 3] and mask.all()


def test_dataset_refuses_to_run_without_a_gpu():
    from llm_quest_amd.dataset import MultimodalDataset

    class Tok:
        eos_token, eos_token_id, pad_token = "<e>", 1, None

        def __call__(self, text):
            return {"input_ids": [2, 3, 1]}

    ds = MultimodalDataset([{"image": np.zeros((4, 4, 3), dtype=np.uint8), "caption_0": "a"}], Tok(), image_size=8, max_caption_len=4)
    assert len(ds) == 1 and ds.tokenizer.pad_token == "<e>"
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            ds[0]
