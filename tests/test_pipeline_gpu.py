"""Input pipeline on HIP kernels vs the oracle (pinned to Pillow): resize + ToTensor + Normalize bit-exact,
pad / truncate / mask exact, the prefetching batch iterator == per-item results."""

import numpy as np
import pytest
import torch

from oracle import pipeline as P

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("needs an MI355X")


class Tok:
    """Minimal tokenizer with the Hugging Face attributes the dataset uses: bytes as ids, eos = 0."""
    eos_token, eos_token_id, pad_token = "\x00", 0, None

    def __call__(self, text):
        return {"input_ids": [b for b in text.encode()]}


def _records(n, seed=3):
    rng = np.random.default_rng(seed)
    recs = []
    for i in range(n):
        h, w = int(rng.integers(20, 400)), int(rng.integers(20, 400))
        recs.append({"image": rng.integers(0, 256, (h, w, 3), dtype=np.uint8), "caption_0": "caption " * int(rng.integers(0, 12)) + str(i)})
    recs[1]["image"] = rng.integers(0, 256, (224, 224, 3), dtype=np.uint8)  # no resize at all
    recs[2]["image"] = rng.integers(0, 256, (224, 100, 3), dtype=np.uint8)  # one pass only
    return recs


@pytest.mark.parametrize("standardize", [True, False])
def test_items_match_the_oracle_bit_for_bit(golden, standardize):
    from llm_quest_amd.dataset import MultimodalDataset

    recs = _records(6)
    ds = MultimodalDataset(recs, Tok(), image_size=224, max_caption_len=40, standardize=standardize)
    for i, r in enumerate(recs):
        item = ds[i]
        want = P.image_transform(r["image"], 224, standardize)
        assert item["image"].dtype == torch.float32 and torch.equal(item["image"].cpu(), want), i
        ids, mask = P.pad_caption(Tok()(r["caption_0"] + "\x00")["input_ids"], 40, 0)
        assert item["input_ids"].dtype == torch.int64 and torch.equal(item["input_ids"].cpu(), ids)
        assert item["attention_mask"].dtype == torch.bool and torch.equal(item["attention_mask"].cpu(), mask)
    # committed Pillow vectors through the kernels
    t = golden("pipeline")
    from llm_quest_amd.dataset import _DeviceTables, image_transform_into

    tabs = _DeviceTables(torch.device("cuda"))
    for name in ("down", "up", "one_pass", "to224"):
        src, want = t[f"resize.{name}.in"], t[f"resize.{name}.out"]
        s_ = want.shape[0]
        out = torch.empty(3, s_, s_, device="cuda")
        image_transform_into(src.cuda(), out, tabs, s_, None, None)
        assert torch.equal(out.cpu(), want.permute(2, 0, 1).float().div(255)), name


def test_prefetched_batches_equal_items():
    from llm_quest_amd.dataset import MultimodalDataset

    recs = _records(11)
    ds = MultimodalDataset(recs, Tok(), image_size=64, max_caption_len=24)
    got = list(ds.batches(4))
    assert [b["image"].shape[0] for b in got] == [4, 4, 3]
    assert len(list(ds.batches(4, drop_last=True))) == 2
    k = 0
    for b in got:
        assert b["image"].is_cuda and b["input_ids"].shape == (b["image"].shape[0], 24) and b["attention_mask"].dtype == torch.bool
        for j in range(b["image"].shape[0]):
            item = ds[k]
            assert torch.equal(b["image"][j], item["image"]) and torch.equal(b["input_ids"][j], item["input_ids"])
            assert torch.equal(b["attention_mask"][j], item["attention_mask"])
            k += 1
    assert k == 11
