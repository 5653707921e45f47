"""The oracle's Philox4x32-10 (oracle/dropout.py) against the published known-answer vectors of the Random123 distribution
(kat_vectors, "philox4x32 10" rows): counters / keys of all zeros, all ones, and the digits of pi."""

import numpy as np

from oracle import dropout as OD


def test_philox4x32_10_known_answers():
    u = np.uint32
    kat = [
        ((0, 0, 0, 0), (0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF,) * 4, (0xFFFFFFFF, 0xFFFFFFFF), (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0), (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    for ctr, key, want in kat:
        got = OD.philox4x32_10(*(u(c) for c in ctr), *key)
        assert tuple(int(g) for g in got) == want


def test_multipliers_are_reproducible_and_shaped():
    a = OD.elementwise_multiplier((3, 5, 7), 0.25, 11, 2)
    assert a.shape == (3, 5, 7) and set(a.unique().tolist()) <= {0.0, OD.inv_keep(0.25)}
    assert (a == OD.elementwise_multiplier((3, 5, 7), 0.25, 11, 2)).all() and not (a == OD.elementwise_multiplier((3, 5, 7), 0.25, 11, 3)).all()
    m = OD.attention_multiplier(2, 2, 9, 0.5, 1, 0)
    assert m.shape == (2, 2, 9, 9) and 0.3 < float((m > 0).float().mean()) < 0.7
    assert OD.threshold(0.0) == 0 and OD.threshold(0.5) == 1 << 31


def test_dropout_stream_leaves_the_cpu_generator_alone_and_follows_the_seed():
    """``nn.Dropout`` on a device tensor consumes the DEVICE generator, never torch's default CPU generator; the (seed, offset) pairs of the
    HIP dropout sites must do the same (llm_quest_amd/rng.py).  Without a device the offsets come from the module's own generator: a new
    ``torch.manual_seed`` restarts the stream, ``get_state`` / ``set_state`` resume it.  (On the GPU the pairs follow the device generator,
    including the same-seed restart: tests/test_dropout_gpu.py.)"""
    import torch

    from llm_quest_amd import rng

    rng.follow_torch()
    torch.manual_seed(1234)
    before = torch.get_rng_state().clone()
    a = [rng.draw() for _ in range(4)]
    assert torch.equal(before, torch.get_rng_state())  # shuffles / torch.rand after a dropout-on run see the generator a dropout-off run sees
    assert len({o for _, o in a}) == 4 and all(s == 1234 for s, _ in a)
    state = rng.get_state()
    c = [rng.draw() for _ in range(2)]
    rng.set_state(state)
    assert [rng.draw() for _ in range(2)] == c
    torch.manual_seed(1235)
    b = rng.draw()
    assert b[0] == 1235 and b != a[0]
    torch.manual_seed(1234)
    assert [rng.draw() for _ in range(4)] == a  # a new seed re-seeds the stream: back to 1234 restarts it
    rng.manual(7, 5)
    assert [rng.draw(), rng.draw()] == [(7, 5), (7, 6)]
    rng.follow_torch()
