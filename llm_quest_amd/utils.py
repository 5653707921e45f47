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
