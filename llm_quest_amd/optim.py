"""AdamW on parameter arenas (optimizer / clip on device).

The reference's step is ``clip_grad_norm_(params, 1.0)`` followed by ``torch.optim.AdamW.step()`` (engine.py:441-450): a
foreach chain over ~310 tensors.  Here the parameters of a transformer block are one flat buffer (llm_quest_amd/arena.py), so
the whole step is: one squared-norm launch per buffer into a device scalar, then ONE fused clip + AdamW launch per buffer --
~60 launches, no host synchronisation, the clip costs no memory pass of its own.  Moments are fp32 regardless of the parameter
dtype (torch keeps bf16 moments for bf16 parameters; fp32 is the more accurate choice and the arenas make it cheap).

``ArenaAdamW`` is a ``torch.optim.Optimizer`` (``param_groups`` / ``zero_grad`` / the reference's ``LearningRateScheduler`` work
unchanged); per-group ``lr`` and ``weight_decay`` are honoured per buffer, so put parameters that need different settings in
different modules / groups.

``torch.optim.AdamW`` semantics that the flat buffers must not blur: a parameter with ``requires_grad=False`` or without a gradient
this step is SKIPPED -- no weight decay, no moment decay.  So an arena is updated as runs of consecutive parameters that are
trainable and have a gradient (one launch per run; the usual case -- everything trained -- is one run = the whole arena).
``state_dict()`` / ``load_state_dict()`` carry the fp32 moments and the step count (under the extra key ``"arena"``: torch would
cast per-parameter state to the parameter dtype, bf16, on load).
"""

import torch

from . import kernels as K


class ArenaAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, max_grad_norm=None):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.max_grad_norm = max_grad_norm
        self._buffers = None  # [(group, param_flat, grad_flat_getter, exp_avg, exp_avg_sq)]
        self._step = 0
        self._pending_state = None  # moments from load_state_dict, adopted when the buffers are planned

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
        if self._pending_state is not None:
            if len(self._pending_state) != len(plan) or any(st["exp_avg"].numel() != e["exp_avg"].numel() for st, e in zip(self._pending_state, plan)):
                raise RuntimeError("ArenaAdamW.load_state_dict: the saved moments do not match this model's parameter buffers")
            for st, e in zip(self._pending_state, plan):
                e["exp_avg"].copy_(st["exp_avg"])
                e["exp_avg_sq"].copy_(st["exp_avg_sq"])
            self._pending_state = None
        return plan

    @staticmethod
    def _runs(arena):
        """[(start, end)) element ranges of the arena that torch.optim.AdamW would update: consecutive parameters that are trainable
        and have a gradient this step (alignment padding between two such parameters rides along: it is zero and stays zero)."""
        runs, start, end = [], None, None
        for p, off in zip(arena.params, arena.offsets):
            live = p.requires_grad and p.grad is not None
            if live:
                if start is None:
                    start = off
                end = off + p.numel()
            elif start is not None:
                runs.append((start, end))
                start = None
        if start is not None:
            runs.append((start, end))
        return runs

    # ------------------------------------------------------------------------------------------------------------------- checkpointing
    def state_dict(self):
        sd = super().state_dict()
        sd["arena"] = {"step": self._step, "buffers": [{"exp_avg": e["exp_avg"], "exp_avg_sq": e["exp_avg_sq"]} for e in (self._buffers or [])]}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        arena = state_dict.pop("arena", None)
        super().load_state_dict(state_dict)
        if arena is not None:
            self._step = int(arena["step"])
            self._pending_state = arena["buffers"] or None
            self._buffers = None

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
        views = []  # (entry, parameter slice, gradient slice, first-moment slice, second-moment slice)
        for e in self._buffers:
            if e["arena"] is not None:
                ar = e["arena"]
                for a, b in self._runs(ar):
                    views.append((e, ar.data[a:b], ar.grad[a:b], e["exp_avg"][a:b], e["exp_avg_sq"][a:b]))
            elif e["param"].grad is not None:
                g = e["param"].grad
                views.append((e, e["param"].data.view(-1), (g if g.is_contiguous() else g.contiguous()).view(-1), e["exp_avg"], e["exp_avg_sq"]))
        sumsq = None
        if self.max_grad_norm is not None and views:
            sumsq = torch.zeros(1, dtype=torch.float32, device=views[0][1].device)
            for v in views:
                K.sumsq_into(v[2], sumsq)
        for e, pflat, gflat, m1, m2 in views:
            grp = e["group"]
            K.adamw_(pflat, gflat, m1, m2, self._step, grp["lr"], grp["betas"], grp["eps"], grp["weight_decay"],
                     sumsq=sumsq, max_norm=self.max_grad_norm or 0.0)
        return None if sumsq is None else sumsq.sqrt()
