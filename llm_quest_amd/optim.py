"""AdamW on parameter arenas (optimizer / clip on device).

The reference's step is ``clip_grad_norm_(params, 1.0)`` followed by ``torch.optim.AdamW.step()`` (engine.py:441-450): a
foreach chain over ~310 tensors.  Here the parameters of a transformer block are one flat buffer (llm_quest_amd/arena.py), so
the whole step is: one squared-norm launch per buffer into a device scalar, then ONE fused clip + AdamW launch per buffer --
~60 launches, no host synchronisation, the clip costs no memory pass of its own.  Moments are fp32 regardless of the parameter
dtype (torch keeps bf16 moments for bf16 parameters; fp32 is the more accurate choice and the arenas make it cheap).

``ArenaAdamW`` is a ``torch.optim.Optimizer`` (``param_groups`` / ``zero_grad`` / the reference's ``LearningRateScheduler`` work
unchanged); per-group ``lr`` and ``weight_decay`` are honoured per buffer, so put parameters that need different settings in
different modules / groups.
"""

import torch

from . import kernels as K


class ArenaAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self._buffers = None  # [(group, param_flat, grad_flat_getter, exp_avg, exp_avg_sq)]
        self._step = 0

    # -------------------------------------------------------------------------------------------------------------------
    def _plan(self):
        """Group every parameter with the arena that owns it (one buffer), the rest become one buffer each."""
        plan, seen = [], set()
        for group in self.param_groups:
            for p in group["params"]:
                if not p.requires_grad:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("ArenaAdamW: parameters must live on the GPU (there is no CPU fallback)")
                arena = self._arena_of(p)
                if arena is not None:
                    if id(arena) in seen:
                        continue
                    seen.add(id(arena))
                    in_group = {id(q) for q in group["params"]}
                    if any(q.requires_grad and id(q) not in in_group for q in arena.params):
                        raise RuntimeError("ArenaAdamW: the parameters of one arena must be in one param group")
                    plan.append(dict(group=group, arena=arena, param=None))
                else:
                    if not p.is_contiguous():
                        raise RuntimeError("ArenaAdamW: stand-alone parameters must be contiguous")
                    plan.append(dict(group=group, arena=None, param=p))
        for e in plan:
            n = e["arena"].data.numel() if e["arena"] is not None else e["param"].numel()
            dev = e["arena"].data.device if e["arena"] is not None else e["param"].device
            e["exp_avg"] = torch.zeros(n, dtype=torch.float32, device=dev)
            e["exp_avg_sq"] = torch.zeros(n, dtype=torch.float32, device=dev)
        return plan

    def _arena_of(self, p):
        for arena in self._known_arenas:
            d = arena.data
            if d is not None and d.data_ptr() <= p.data_ptr() < d.data_ptr() + d.numel() * d.element_size():
                return arena
        return None

    def attach(self, *modules):
        """Tell the optimizer which modules own arenas (anything with ``_arena`` or an ``arenas()`` method).  Parameters found in
        none of them are updated as stand-alone buffers."""
        known = []
        for m in modules:
            if hasattr(m, "arenas"):
                known += list(m.arenas())
            for sub in m.modules():
                a = getattr(sub, "_arena", None)
                if a is not None and all(a is not b for b in known):
                    known.append(a)
        self._known_arenas = known
        self._buffers = None
        return self

    _known_arenas = ()

    # -------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step(self, closure=None):
        if closure is not None:
            raise NotImplementedError("ArenaAdamW.step: closures are not supported")
        if self._buffers is None or any(e["arena"] is not None and e["arena"].data.numel() != e["exp_avg"].numel() for e in self._buffers):
            self._buffers = self._plan()
        self._step += 1
        views = []
        for e in self._buffers:
            if e["arena"] is not None:
                e["arena"].untouched_to_zero()
                views.append((e, e["arena"].data, e["arena"].grad))
            elif e["param"].grad is not None:
                g = e["param"].grad
                views.append((e, e["param"].data.view(-1), (g if g.is_contiguous() else g.contiguous()).view(-1)))
        sumsq = None
        if self.max_grad_norm is not None and views:
            sumsq = torch.zeros(1, dtype=torch.float32, device=views[0][1].device)
            for _, _, g in views:
                K.sumsq_into(g, sumsq)
        for e, pflat, gflat in views:
            grp = e["group"]
            K.adamw_(pflat, gflat, e["exp_avg"], e["exp_avg_sq"], self._step, grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"],
                     sumsq=sumsq, max_norm=self.max_grad_norm or 0.0)
        return None if sumsq is None else sumsq.sqrt()
