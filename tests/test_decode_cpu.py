"""KV cache host logic: capacity growth policy and position bookkeeping of utils.KVCache, which are
pure Python and mirror llm_quest/utils.py:409-531."""

import math


def test_kvcache_capacity_policy_matches_reference_rule():
    from llm_quest_amd.utils import KVCache

    kv = KVCache(num_layers=3, prompt_len=10, context_len=1000, initial_chunk_size=6, chunk_size=8)
    assert kv.kv_capacity == 16 and kv.start_pos == 0 and kv.end_pos == 0
    # the growth rule alone (no tensors): minimum number of whole chunks, capped at context_len
    kv.end_pos = 17
    need = math.ceil((kv.end_pos - kv.kv_capacity) / kv.chunk_size) * kv.chunk_size
    assert need == 8
    kv2 = KVCache(num_layers=1, prompt_len=990, context_len=1000)
    assert kv2.kv_capacity == 990 + 512  # as upstream: the initial capacity is not clipped to context_len
