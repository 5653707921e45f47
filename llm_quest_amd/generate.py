"""Text generation loops -- API of ``llm_quest/generate.py`` for the paths on SURVEY.md section 8 row f4: ``generate_loop``
(full re-forward per token), ``generate_loop_kv_cache`` (prefill + one-token decode steps through ``KVCache``) and ``sampling``.

Greedy sampling (``temp == 0``, the default) is a HIP kernel (first index of the row maximum).  Stochastic sampling (temperature /
top-k / top-p / min-p) is control logic on a (batch, vocab) tensor per generated token, outside the measured path: it is restated
here with torch device ops so the loops are complete, and says so.
"""

import torch

from . import ops_decode
from .utils import KVCache


def _top_k(probs, k):
    top, idx = torch.topk(probs, k)
    return torch.zeros_like(probs).scatter_(-1, idx, top)


def _top_p(probs, p, top_k=None):
    if top_k:
        kth = torch.topk(probs, top_k)[0][..., -1].unsqueeze(-1)
        probs = probs.masked_fill(probs < kth, 0.0)
    sp, idx = torch.sort(probs, dim=-1, descending=True)
    mask = torch.cumsum(sp, dim=-1) > p
    mask[..., 1:] = mask[..., :-1].clone()
    mask[..., 0] = False
    return torch.zeros_like(probs).scatter_(-1, idx, sp.masked_fill(mask, 0.0))


def _min_p(probs, min_p, min_tokens_to_keep=1):
    remove = probs < min_p * torch.amax(probs, dim=-1, keepdim=True)
    keep_idx = torch.topk(probs, min_tokens_to_keep)[1]
    remove.scatter_(-1, keep_idx, False)
    return probs.masked_fill(remove, 0.0)


def sampling(logits, top_k=None, top_p=None, min_p=None, temp=0.0):
    """logits (b, v) -> next token ids (b, 1) (reference: generate.py:472-514)."""
    assert top_p is None or min_p is None, "Cannot use top_p and min_p together"
    if temp == 0.0:
        return ops_decode.argmax_rows(logits if logits.stride(-1) == 1 else logits.contiguous()).unsqueeze(-1)
    probs = torch.softmax(logits.float() / temp, dim=-1)
    if min_p:
        probs = _min_p(probs, min_p, 1 if top_k is None else top_k)
    elif top_p:
        probs = _top_p(probs, top_p, top_k)
    elif top_k:
        probs = _top_k(probs, top_k)
    probs = probs / probs.sum(dim=-1, keepdim=True)
    return torch.multinomial(probs, num_samples=1)


def generate_loop(input_tensor, model, max_gen, context_length, top_k=None, top_p=None, min_p=None, temp=0.0, eos_ids=None, device=torch.device("cuda")):
    """Re-runs the model on the growing sequence for every token (reference: generate.py:29-94)."""
    input_tensor = input_tensor.to(device)
    eos = None
    if eos_ids is not None:
        eos = torch.tensor(eos_ids if isinstance(eos_ids, list) else [eos_ids], device=device, dtype=torch.long)
    for _ in range(max_gen):
        with torch.inference_mode():
            logits = model(input_tensor[:, -context_length:])[:, -1, :]
        next_token = sampling(logits, top_k, top_p, min_p, temp)
        if eos is not None and torch.isin(next_token, eos).any():
            break
        input_tensor = torch.cat((input_tensor, next_token), dim=-1)
    return input_tensor


def generate_loop_kv_cache(input_tensor, model, max_gen, context_length, top_k=None, top_p=None, min_p=None, temp=0.0, eos_ids=None,
                           device=torch.device("cuda"), use_graph=True):
    """Prefill once, then one-token decode steps through the KV cache (reference: generate.py:97-151).  Greedy decoding
    (``temp == 0``) replays one captured hipGraph per token (``use_graph``); sampling runs the steps eagerly."""
    if use_graph and temp == 0.0 and max_gen > 0 and hasattr(model, "trf_blocks") and input_tensor.shape[-1] + max_gen <= context_length:
        return _generate_greedy_graph(input_tensor, model, max_gen, context_length, eos_ids, device)
    token_ids = []
    kv_cache = KVCache(num_layers=len(model.trf_blocks), prompt_len=input_tensor.shape[-1], context_len=context_length)
    input_tensor = input_tensor.to(device)
    eos = None
    if eos_ids is not None:
        eos = torch.tensor(eos_ids if isinstance(eos_ids, list) else [eos_ids], device=device, dtype=torch.long)
    trunc_input = input_tensor[:, -context_length:]
    next_position_id = torch.tensor([[trunc_input.shape[-1]]], dtype=torch.long, device=device)
    with torch.inference_mode():
        logits = model(trunc_input, kv_cache=kv_cache)[:, -1, :]
        for _ in range(max_gen):
            next_token = sampling(logits, top_k, top_p, min_p, temp)
            if eos is not None and torch.isin(next_token, eos).any():
                break
            token_ids.append(next_token)
            logits = model(next_token, kv_cache=kv_cache, position_ids=next_position_id).squeeze(1)
            next_position_id += 1
    return torch.cat([input_tensor] + token_ids, dim=-1)


def _generate_greedy_graph(input_tensor, model, max_gen, context_length, eos_ids, device):
    token_ids = []
    kv_cache = KVCache(num_layers=len(model.trf_blocks), prompt_len=input_tensor.shape[-1], context_len=context_length)
    input_tensor = input_tensor.to(device)
    eos = None
    if eos_ids is not None:
        eos = torch.tensor(eos_ids if isinstance(eos_ids, list) else [eos_ids], device=device, dtype=torch.long)
    with torch.inference_mode():
        logits = model(input_tensor[:, -context_length:], kv_cache=kv_cache)[:, -1, :]
        next_token = sampling(logits)
        dec = None
        try:
            for _ in range(max_gen):
                if eos is not None and torch.isin(next_token, eos).any():
                    break
                token_ids.append(next_token)
                if len(token_ids) == max_gen:
                    break
                if dec is None:
                    dec = ops_decode.GraphDecoder(model, kv_cache, next_token, max_gen)
                next_token = dec.step()
        finally:
            if dec is not None:
                dec.close()
    return torch.cat([input_tensor] + token_ids, dim=-1)


def _batched_kv_loop(input_tensor, model, max_gen, context_length, top_k, top_p, min_p, temp, eos_ids, pad_id, device, attention_mask,
                     first_position_ids, next_pos_ids, pick_first_logits):
    """Shared body of the two padded, batched KV-cache loops: prefill, then one token per unfinished sequence; finished sequences keep
    receiving ``pad_id`` and are masked out of later attention (reference: generate.py:252-365, 368-469)."""
    input_tensor = input_tensor.to(device)
    attention_mask = attention_mask.bool().to(device)
    eos = torch.tensor(eos_ids if isinstance(eos_ids, list) else [eos_ids], device=device, dtype=torch.long)
    pad = torch.tensor(pad_id, device=device, dtype=torch.long)
    finished = torch.zeros(input_tensor.shape[0], dtype=torch.bool, device=device)
    kv_cache = KVCache(num_layers=len(model.trf_blocks), prompt_len=input_tensor.shape[-1], context_len=context_length)
    generated = []
    with torch.inference_mode():
        logits = pick_first_logits(model(input_tensor, attn_mask=attention_mask, kv_cache=kv_cache, position_ids=first_position_ids))
        for i in range(max_gen):
            next_token = torch.where(finished.unsqueeze(-1), pad, sampling(logits, top_k, top_p, min_p, temp))
            generated.append(next_token)
            finished |= torch.isin(next_token.squeeze(1), eos)
            attention_mask = torch.cat([attention_mask, (~finished).unsqueeze(-1)], dim=-1)
            if finished.all():
                break
            if i < max_gen - 1:
                logits = model(next_token, attn_mask=attention_mask, kv_cache=kv_cache, position_ids=next_pos_ids).squeeze(1)
            next_pos_ids = next_pos_ids + 1
    return torch.cat([input_tensor] + generated, dim=1)


def generate_batched_loop_kv_cache(input_tensor, model, max_gen, context_length, top_k=None, top_p=None, min_p=None, temp=0.0, eos_ids=50256,
                                   pad_id=50256, device=torch.device("cuda"), last_real=None, *, attention_mask):
    """RIGHT-padded prompts: the first logits come from each sequence's last real token, rotary positions continue from its real
    length (reference: generate.py:252-365)."""
    attention_mask = attention_mask.bool().to(device)
    if last_real is not None:
        last_real = last_real.to(device)
        next_pos = last_real.unsqueeze(-1) + 1
    else:
        next_pos = attention_mask.sum(dim=-1, keepdim=True)
        last_real = next_pos.squeeze(-1) - 1
    rows = torch.arange(input_tensor.shape[0], device=device)
    return _batched_kv_loop(input_tensor, model, max_gen, context_length, top_k, top_p, min_p, temp, eos_ids, pad_id, device, attention_mask,
                            None, next_pos, lambda lg: lg[rows, last_real, :])


def generate_batched_loop_kv_cache_left_pad(input_tensor, model, max_gen, context_length, top_k=None, top_p=None, min_p=None, temp=0.0,
                                            eos_ids=50256, pad_id=50256, device=torch.device("cuda"), *, attention_mask):
    """LEFT-padded prompts: positions count real tokens only (pads sit at position 0), every sequence ends in the last column
    (reference: generate.py:368-469)."""
    attention_mask = attention_mask.bool().to(device)
    position_ids = (attention_mask.cumsum(dim=-1) - 1).masked_fill(~attention_mask, 0)
    next_pos = attention_mask.sum(dim=-1, keepdim=True)
    return _batched_kv_loop(input_tensor, model, max_gen, context_length, top_k, top_p, min_p, temp, eos_ids, pad_id, device, attention_mask,
                            position_ids, next_pos, lambda lg: lg[:, -1, :])
