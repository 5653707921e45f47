"""Flat HBM arenas for parameters and their gradients.

MI355X-first memory layout: all parameters of one transformer block live back to back in ONE device buffer (and their
gradients in a second, same-shaped buffer).  The individual ``nn.Parameter`` objects keep the reference's names and shapes
(state_dict compatibility) but their ``.data`` are views into the arena, which buys:

  * fused projections without copies -- ``w_queries|w_keys|w_values`` (and ``lin1|lin_gate``) are adjacent row blocks of
    one [N_total, K] matrix, so QKV / gate-up run as single GEMMs and single wgrad GEMMs;
  * one contiguous gradient bucket per block for the RCCL all-reduce (no flatten/unflatten copies);
  * wgrad kernels write straight into the bucket (``p.grad`` is a view of it).

``nn.Module.to()`` / ``load_state_dict`` can replace or refill ``p.data``; ``ensure()`` is called at the top of every
forward and rebuilds the arena when the views no longer point into it.
"""

import torch


class ParamArena:
    def __init__(self, named_params):
        self.names = [n for n, _ in named_params]
        self.params = [p for _, p in named_params]
        self.data = None
        self.grad = None
        self.offsets = []
        self._ptrs = None

    # ------------------------------------------------------------------ storage
    def _layout(self):
        offs, off = [], 0
        for p in self.params:
            offs.append(off)
            off += (p.numel() + 7) // 8 * 8  # keep every view 16-byte aligned
        return offs, off

    def ensure(self):
        """(Re)build the arena if any parameter's storage is not the expected view.  Cheap when nothing moved."""
        ptrs = tuple(p.data_ptr() for p in self.params)
        if self.data is not None and ptrs == self._ptrs and self.params[0].device == self.data.device:
            return False
        p0 = self.params[0]
        for p in self.params:
            if p.dtype != p0.dtype or p.device != p0.device:
                raise RuntimeError("ParamArena: parameters of one arena must share dtype and device")
        self.offsets, total = self._layout()
        data = torch.empty(total, dtype=p0.dtype, device=p0.device)
        with torch.no_grad():
            for p, off in zip(self.params, self.offsets):
                view = data[off : off + p.numel()].view(p.shape)
                view.copy_(p.data)
                p.data = view
                p.grad = None
        self.data = data
        self.grad = torch.zeros(total, dtype=p0.dtype, device=p0.device)
        self._fresh = [True] * len(self.params)
        self._ptrs = tuple(p.data_ptr() for p in self.params)
        return True

    def index(self, p):
        for i, q in enumerate(self.params):
            if q is p:
                return i
        raise KeyError("parameter not in arena")

    def fused(self, first, last):
        """2-D view [sum(rows), K] over the consecutive 2-D parameters first..last (inclusive)."""
        i, j = self.index(first), self.index(last)
        k = first.shape[1]
        rows = 0
        for t in range(i, j + 1):
            p = self.params[t]
            if p.dim() != 2 or p.shape[1] != k or self.offsets[t] != self.offsets[i] + rows * k:
                raise RuntimeError("ParamArena.fused: parameters are not adjacent row blocks")
            rows += p.shape[0]
        return self.data[self.offsets[i] : self.offsets[i] + rows * k].view(rows, k)

    # ------------------------------------------------------------------ gradients
    def grad_view(self, i):
        p = self.params[i]
        off = self.offsets[i]
        return self.grad[off : off + p.numel()].view(p.shape)

    def grad_target(self, first, last=None):
        """Destination for a wgrad kernel covering parameters first..last: returns (view, accumulate).

        Attaches ``p.grad`` to the arena when it is None (after ``zero_grad(set_to_none=True)``); ``accumulate`` is False
        only if every covered gradient is stale and may simply be overwritten.
        """
        i = self.index(first)
        j = i if last is None else self.index(last)
        states = []
        for t in range(i, j + 1):
            p = self.params[t]
            gv = self.grad_view(t)
            if p.grad is None:
                p.grad = gv
                self._fresh[t] = True
            elif p.grad.data_ptr() != gv.data_ptr():
                raise RuntimeError(
                    f"gradient of '{self.names[t]}' was replaced by a foreign tensor; use zero_grad() or leave .grad alone"
                )
            states.append(self._fresh[t])
        if any(states) and not all(states):
            for t in range(i, j + 1):  # mixed: zero the stale ones, then accumulate everywhere
                if self._fresh[t]:
                    self.grad_view(t).zero_()
        accumulate = not all(states)
        for t in range(i, j + 1):
            self._fresh[t] = False
        start = self.offsets[i]
        end = self.offsets[j] + self.params[j].numel()
        flat = self.grad[start:end]
        if last is None or i == j:
            return flat.view(first.shape), accumulate
        k = first.shape[1]
        return flat.view(-1, k), accumulate

    def untouched_to_zero(self):
        """Parameters whose gradient was not produced this backward get a zero gradient (attached, fresh cleared)."""
        for t, p in enumerate(self.params):
            if p.requires_grad and p.grad is None:
                gv = self.grad_view(t)
                gv.zero_()
                p.grad = gv
                self._fresh[t] = False

    def trainable(self):
        return any(p.requires_grad for p in self.params)
