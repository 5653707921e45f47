"""``KVCache`` of the reference (llm_quest/utils.py:409-531) with an MI355X-first layout.

Same constructor, growth policy (``prompt_len + initial_chunk_size``, then whole ``chunk_size`` steps up to ``context_len``) and
bookkeeping (``start_pos`` / ``end_pos`` advance after the last layer) as upstream, but the cached keys / values are kept
TOKEN-MAJOR -- ``[batch, capacity, kv_heads * head_dim]`` bf16, exactly the rows the fused QKV projection and the QK-norm + RoPE
kernel produce -- so appending is a strided row copy and the decode attention kernel streams contiguous 256-byte head rows;
upstream's ``(batch, heads, capacity, head_dim)`` would need a transpose per token.  ``get_updated_cache`` keeps the reference's
call signature on ``(b, heads, s, d)`` tensors for code that uses the cache directly.
"""

import math

import torch

from . import _lib as L
from . import kernels as K


class KVCache:
    def __init__(self, num_layers, prompt_len, context_len, initial_chunk_size=512, chunk_size=256):
        self.num_layers = num_layers
        self.prompt_len = prompt_len
        self.context_len = context_len
        self.chunk_size = chunk_size
        self.kv_capacity = self.prompt_len + initial_chunk_size
        self.keys_cache = []
        self.values_cache = []
        self.start_pos = 0
        self.end_pos = 0

    # ------------------------------------------------------------------ storage
    def _initialize(self, batch_size, width, device, dtype):
        for _ in range(self.num_layers):
            self.keys_cache.append(torch.zeros(batch_size, self.kv_capacity, width, device=device, dtype=dtype))
            self.values_cache.append(torch.zeros(batch_size, self.kv_capacity, width, device=device, dtype=dtype))

    def _grow_kv_capacity(self, layer_idx):
        if self.kv_capacity < self.end_pos:
            if self.end_pos < self.context_len:
                self.kv_capacity += math.ceil((self.end_pos - self.kv_capacity) / self.chunk_size) * self.chunk_size
            else:
                self.kv_capacity = self.context_len
        for cache in (self.keys_cache, self.values_cache):
            old = cache[layer_idx]
            new = torch.empty(old.shape[0], self.kv_capacity, old.shape[2], device=old.device, dtype=old.dtype)
            if self.start_pos:
                for b in range(old.shape[0]):  # bit-exact strided row copies
                    K.copy2d(old[b, : self.start_pos], new[b, : self.start_pos])
            cache[layer_idx] = new

    def reserve(self, capacity):
        """Grow every layer's buffers to at least ``capacity`` rows now (contents kept), so that later appends never reallocate --
        required before a decode step is captured in a hipGraph."""
        capacity = min(int(capacity), self.context_len)
        if not self.keys_cache:
            self.kv_capacity = max(self.kv_capacity, capacity)
            return
        if capacity <= min(c.shape[1] for c in self.keys_cache):
            return
        self.kv_capacity = max(self.kv_capacity, capacity)
        saved_end = self.end_pos
        self.end_pos = self.kv_capacity
        for layer_idx in range(self.num_layers):
            if self.keys_cache[layer_idx].shape[1] < self.kv_capacity:
                self._grow_kv_capacity(layer_idx)
        self.end_pos = saved_end

    def append_rows(self, k_rows, v_rows, layer_idx, batch_size, new_seq_len):
        """k_rows / v_rows: token-major [batch*new_seq_len, kv_heads*head_dim] (row-strided views allowed).  Returns
        (k_cache [batch, capacity, width], v_cache, end_pos) with the new rows in place at [start_pos, end_pos)."""
        L.require_gpu(k_rows, v_rows)
        width = k_rows.shape[1]
        if not self.keys_cache:
            self.batch_size, self.device, self.dtype = batch_size, k_rows.device, k_rows.dtype
            self._initialize(batch_size, width, k_rows.device, k_rows.dtype)
        self.end_pos = self.start_pos + new_seq_len
        if self.end_pos > self.context_len:
            raise ValueError(f"KVCache: sequence length {self.end_pos} exceeds context_len {self.context_len}")
        if self.end_pos > self.keys_cache[layer_idx].shape[1]:
            self._grow_kv_capacity(layer_idx)
        kc, vc = self.keys_cache[layer_idx], self.values_cache[layer_idx]
        for b in range(batch_size):
            K.copy2d(k_rows[b * new_seq_len : (b + 1) * new_seq_len], kc[b, self.start_pos : self.end_pos])
            K.copy2d(v_rows[b * new_seq_len : (b + 1) * new_seq_len], vc[b, self.start_pos : self.end_pos])
        end = self.end_pos
        if layer_idx == self.num_layers - 1:
            self.start_pos += new_seq_len
        return kc, vc, end

    # ------------------------------------------------------------------ reference signature
    def get_updated_cache(self, keys, values, layer_idx):
        """keys / values (batch, heads, new_seq_len, head_dim) -> the cached (batch, heads, end_pos, head_dim) tensors (views of
        the token-major storage), as utils.py:496-531."""
        b, h, s, d = keys.shape
        self.num_heads, self.head_dim = h, d
        tm = lambda t: t.permute(0, 2, 1, 3).reshape(b * s, h * d).contiguous()
        kc, vc, end = self.append_rows(tm(keys), tm(values), layer_idx, b, s)
        hm = lambda c: c[:, :end].view(b, end, h, d).permute(0, 2, 1, 3)
        return hm(kc), hm(vc)


class Qwen3_5Cache:
    """Hybrid cache of the Qwen3.5 text stack (reference: llm_quest/utils.py:535-624): a ``KVCache`` for the full-attention layers
    (indexed by their position among the full-attention layers) and, per linear-attention layer, two fixed-size states --
    ``conv_states[i]`` bf16 [batch, kernel_size, fused_dim] (TOKEN-MAJOR here; upstream keeps (batch, fused_dim, kernel_size)) and
    ``recurrent_states[i]`` fp32 [batch, value_heads, v_head_dim, qk_head_dim] -- both updated in place by the decode kernels."""

    def __init__(self, n_layers, linear_sdpa_ratio, prompt_len, context_len):
        self.n_layers = n_layers
        self.linear_sdpa_ratio = linear_sdpa_ratio
        self.layer_types = ["full_attention" if (i + 1) % linear_sdpa_ratio == 0 else "linear_attention" for i in range(n_layers)]
        self.full_attn_indices = [i for i, t in enumerate(self.layer_types) if t == "full_attention"]
        self._full_attn_to_kv_idx = {g: i for i, g in enumerate(self.full_attn_indices)}
        self.kv_cache = KVCache(num_layers=len(self.full_attn_indices), prompt_len=prompt_len, context_len=context_len)
        self.conv_states = [None] * n_layers
        self.recurrent_states = [None] * n_layers

    def get_updated_kv_cache(self, keys, values, layer_idx):
        return self.kv_cache.get_updated_cache(keys, values, self._full_attn_to_kv_idx[layer_idx])

    def append_kv_rows(self, k_rows, v_rows, layer_idx, batch_size, new_seq_len):
        return self.kv_cache.append_rows(k_rows, v_rows, self._full_attn_to_kv_idx[layer_idx], batch_size, new_seq_len)

    @property
    def has_previous_state(self):
        return any(s is not None for t, s in zip(self.layer_types, self.conv_states) if t == "linear_attention")

    def get_conv_state(self, layer_idx):
        return self.conv_states[layer_idx]

    def set_conv_state(self, layer_idx, conv_state):
        self.conv_states[layer_idx] = conv_state

    def get_recurrent_state(self, layer_idx):
        return self.recurrent_states[layer_idx]

    def set_recurrent_state(self, layer_idx, recurrent_state):
        self.recurrent_states[layer_idx] = recurrent_state


# ----------------------------------------------------------------------------------------------- checkpoint import (row f2)
# API of the "WEIGHTS LOADING" helpers of llm_quest/utils.py:921-1035.  The build / GPU boxes have no network, so
# ``download_hf_weights`` resolves its argument against the local filesystem (a .safetensors file, a directory of shards with
# or without ``model.safetensors.index.json``) instead of the hub; everything downstream of it behaves as upstream.
def read_local_checkpoint(source):
    """dict of tensors, a ``.safetensors`` file, or a directory of ``*.safetensors`` shards -> one state dict."""
    import glob
    import json
    import os

    if isinstance(source, dict):
        return source
    from safetensors.torch import load_file

    if os.path.isdir(source):
        index = os.path.join(source, "model.safetensors.index.json")
        if os.path.exists(index):
            with open(index) as f:
                names = sorted(set(json.load(f)["weight_map"].values()))
            files = [os.path.join(source, n) for n in names]
        else:
            files = sorted(glob.glob(os.path.join(source, "*.safetensors")))
        if not files:
            raise FileNotFoundError(f"no .safetensors shards under {source}")
    elif os.path.exists(source):
        files = [source]
    else:
        raise FileNotFoundError(
            f"'{source}' is not a local checkpoint.  This package never opens a network connection: fetch the Hugging Face "
            "snapshot yourself and pass its directory (as model_cfg['model_path'] or as source=...)")
    state = {}
    for f in files:
        state.update(load_file(f))
    return state


def resolve_checkpoint(source, model_cfg):
    """``source`` (dict / file / shard directory) or, without it, the local snapshot that ``model_cfg["model_path"]`` names."""
    if source is not None:
        return read_local_checkpoint(source)
    if not model_cfg.get("model_path"):
        raise ValueError("no checkpoint given: pass source=<dict | .safetensors file | shard directory> or set model_cfg['model_path'] "
                         "to a local snapshot (downloading from the Hugging Face hub is not available in this environment)")
    return download_hf_weights(model_cfg["model_path"])


def download_hf_weights(hf_model_name):
    """The reference downloads ``hf_model_name`` from the hub (utils.py:921-953); here the name must be a local snapshot."""
    print(f"Loading {hf_model_name} from local storage...")
    state = read_local_checkpoint(hf_model_name)
    print(f"Successfully loaded weights from {hf_model_name}")
    return state


def _target_name(hf_name, rules):
    """Name a checkpoint tensor gets here: the (pattern, replacement) pairs rewrite substrings one after the other, and a pair
    whose pattern is the COMPLETE original name closes the rewriting as soon as it fired (top-level tensors such as
    ``model.norm.weight`` must not be touched by the per-layer suffix rules that follow)."""
    name, k = hf_name, 0
    while k < len(rules):
        pattern, replacement = rules[k]
        k += 1
        if pattern not in name:
            continue
        name = name.replace(pattern, replacement)
        if pattern == hf_name:
            k = len(rules)
    return name


def convert_weights(hf_state_dict, our_state_dict, remapping_rules, ignored_prefixes=None):
    """Hugging Face tensors under this package's parameter names (reference utils.py:956-1000).  A tensor is kept when its
    target name exists in ``our_state_dict`` with the same shape; the others are reported and left out, tensors under
    ``ignored_prefixes`` silently (counted)."""
    ignored = tuple(ignored_prefixes) if ignored_prefixes else ()
    wanted = {n: w for n, w in hf_state_dict.items() if not (ignored and n.startswith(ignored))}
    plan = {n: _target_name(n, remapping_rules) for n in wanted}
    converted = {}
    for hf_name, ours in plan.items():
        have = our_state_dict.get(ours)
        if have is None:
            print(f"WARNING: No match for HF weight '{hf_name}' → tried '{ours}'")
        elif have.shape != wanted[hf_name].shape:
            print(f"WARNING: Shape mismatch: {ours}: HF {wanted[hf_name].shape} vs Ours {have.shape}")
        else:
            converted[ours] = wanted[hf_name].clone()
    if len(wanted) != len(hf_state_dict):
        print(f"Skipped {len(hf_state_dict) - len(wanted)} weights")
    return converted


def handle_weight_tying(model):
    """After ``load_state_dict``: make ``out_head.weight`` the embedding matrix again when the model ties them
    (reference utils.py:1003-1022)."""
    if not getattr(model, "tie_embeddings", False):
        print("Tie_embeddings=False, skipping weight tying\n")
        return
    emb, head = model.emb_dict.weight, model.out_head.weight
    if emb.shape != head.shape:
        print(f"WARNING: Shape mismatch for weight tying: {emb.shape} vs {head.shape}")
        return
    model.out_head.weight = emb
    print("Weight tied successfully\n" if model.out_head.weight is model.emb_dict.weight else "WARNING: Weight tying failed!\n")


def report_loading_status(model, load_result, converted_weights):
    """What was loaded, what the model still misses, what the checkpoint had in excess (reference utils.py:1025-1035)."""
    print(f"Loaded {len(converted_weights)}/{len(model.state_dict())} weights successfully\n")
    if load_result.missing_keys:
        print(f"Missing keys ({len(load_result.missing_keys)}): {load_result.missing_keys}")
        print("-> out_head is expected here with tie_embeddings=True, and so are the locally rebuilt buffers (mask, cos, sin)\n")
    if load_result.unexpected_keys:
        print(f"Unexpected keys: {load_result.unexpected_keys}")
