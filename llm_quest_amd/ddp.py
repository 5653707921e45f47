"""Data-parallel gradient exchange: one process per GPU, RCCL all-reduce over xGMI overlapped with backward.

The reference has no distributed code at all; BASELINE's north star adds exactly one strategy:
pure data parallelism over one 8 x MI355X node.  Design:

  * every rank holds a full replica and an independent micro-batch shard (weak scaling);
  * gradients already live in flat per-block arenas (arena.py), so a bucket is just ``arena.grad`` -- no flatten copies;
  * the moment a block's backward finishes (``blk._grad_ready`` hook, fired by ops.block_backward) an event is recorded on
    the compute stream and the bucket's all-reduce is enqueued on a dedicated communication stream that waits on it, so
    the exchange of block i runs under the backward of blocks i-1, i-2, ...;
  * ``finish_step`` reduces what cannot be known complete earlier (embedding / LM-head arena) and makes the compute
    stream wait for the communication stream -- the global-norm clip and the optimizer then see averaged gradients;
  * xGMI is point-to-point (7 links x ~153 GB/s per GPU): with ~31 MB bf16 per Qwen3-0.6B block the 30 collectives per
    step are large enough to be link-bound, not launch-bound, and small enough to overlap at block granularity.

Loss weighting: ``vlm_loss`` is a mean over non-ignored tokens per rank; averaging gradients across ranks equals the
single-process global-batch gradient only when every rank has the same number of target tokens (the synthetic all-ones
mask).  For ragged masks multiply the loss by ``sync.loss_weight(n_target_tokens)`` before ``backward()``: the factor is
``n_r * world / sum_r n_r`` (one 1-element all-reduce of the counts, no host sync), so the AVG all-reduce of the buckets
returns ``sum_r grad(sum-loss_r) / sum_r n_r`` -- the gradient of the global-batch mean.

The tied embedding / LM-head matrix (311 MB, the largest bucket) receives the head's weight gradient FIRST in the backward and the embedding's
scatter LAST, so as one bucket it can only be exchanged after the backward, fully exposed (~1.8 ms on xGMI: a ring all-reduce moves
2 (w - 1) / w x 311 MB per rank).  It CAN be split (``early_tail``): the dense part -- head gradient + final norm -- is all-reduced as soon as the
last transformer block's backward has finished, under the 27 blocks still to come; the embedding's part is sparse (one row per token), so at the end
the ranks all-gather their token ids and token-gradient rows and every rank runs the same deterministic segmented sum
(``mi355_embedding_bwd_sorted``) over all ranks' tokens with scale 1 / world -- identical bits on every rank, nothing dense left to exchange.
Whether that pays is a matter of BYTES, decided per step in ``begin_step(embedding_tokens=T)`` (``split_pays``): the exposed all-gather delivers
(w - 1) x T x (row bytes + 8) to every rank against the dense ring's 2 (w - 1) / w x bucket bytes.  At 64 x 512 tokens and 8 ranks that is 0.47 GB
against 0.54 GB (split); at the bench's 160 x 512 it is 1.18 GB against 0.54 GB, so the bucket stays whole and goes out in ``finish_step``; at 2 ranks
and 160 x 512 it is 0.17 GB against 0.31 GB (split).  Without a token count (``begin_step()``) the bucket stays whole.  ``MI355_DDP_SPLIT_TIED`` =
``on`` / ``off`` overrides the rule (A/B runs).  The all-gather needs equally many tokens on every rank: every step that passes a count checks that
with one 16-byte all-reduce (MIN / MAX of T, on a control stream of its own) and raises on EVERY rank instead of hanging in ``all_gather_into_tensor``.
All ranks must either pass a count or pass none.

Bucket groups and GEMM windows (round 5).  The compute stream's large NT GEMMs run on a persistent kernel -- one workgroup per CU for the launch's whole length,
all 256 starting together -- and a collective's channels are workgroups that hold CUs: a persistent launch beside a collective lasts up to twice as long, the per-tile
form of the same kernel loses 17-27 % (tools/contention.py).  So the block buckets go out in GROUPS of ``MI355_DDP_BUCKET_BLOCKS`` (default 7: four groups for the 28
blocks; the collectives of a group run back to back on the communication stream from the moment its last block is complete), and behind each group the compute stream
runs a WINDOW: its next ``MI355_DDP_WINDOW_LAUNCHES`` (default 2) persistent-sized NT launches take the per-tile kernel (``kernels.open_gemm_window``) -- the first two
dgrad GEMMs of the following block, ~2.2 ms against ~1.3 ms of ring all-reduce for 7 x 31.5 MB on eight ranks -- then the compute stream waits for the group's
completion event (a bound on the worst case, not a stall in the normal one) and the persistent kernel is back.  Five windows per step instead of a whole backward on
the slower kernel; ``kernels._WINDOW.stats`` counts both kinds of launch and ``bench.py`` prints them for N > 1.
The window's length is MEASURED, not assumed (round 6): every group's collectives are bracketed by two events on the communication stream, every window by two on the
compute stream; ``begin_step`` reads the pairs that have completed and sets ``window_launches = ceil(slowest group's all-reduce / one windowed launch)`` (1..16), so on
a slower fabric, more ranks or another node count the windows grow with the collectives instead of the compute stream stalling at their end (``window_for``;
``MI355_DDP_WINDOW_LAUNCHES`` pins the length, ``MI355_DDP_WINDOW_WAIT=0`` drops the bound at a window's end altogether).

Gradient accumulation: wrap every micro-step but the last in ``with sync.no_sync():`` -- the hooks then leave the buckets
alone (they keep accumulating locally) and ``finish_step`` is a no-op; the last micro-step exchanges the sums.
"""

import contextlib

import os

import torch
import torch.distributed as dist


_ACTIVE = None  # the GradSync between begin_step() and finish_step(): autograd nodes that split their bucket ask it (ops.EmbeddingFn)


def active():
    return _ACTIVE


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  world_size 1 -> no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        # RCCL's collectives are kernels: each channel is one workgroup on one CU, taken from the compute stream's GEMMs (whose 256x256 tiles run one
        # 8-wave workgroup per CU and size their grids for 256 free CUs: DESIGN.md section 6).  MI355_RCCL_CHANNELS caps / pins that number for A/B
        # runs on an 8-GPU node (it sets NCCL_MIN_NCHANNELS / NCCL_MAX_NCHANNELS, which RCCL reads at communicator creation); unset = RCCL's own choice.
        ch = os.environ.get("MI355_RCCL_CHANNELS")
        if ch:
            os.environ.setdefault("NCCL_MAX_NCHANNELS", ch)
            os.environ.setdefault("NCCL_MIN_NCHANNELS", ch)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"  # "nccl" is RCCL on ROCm
        if backend == "nccl":
            torch.cuda.set_device(local)
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


class GradSync:
    """All-reduce (average) of gradient arenas, overlapped with backward.

    ``owners`` are modules that fire ``_grad_ready(module)`` when their arena's gradients are final for this backward
    (Qwen3 TransformerBlocks, ViTAdapter); ``tail_arenas`` are reduced in ``finish_step`` (embedding / LM-head arena).
    """

    def __init__(self, owners, tail_arenas=(), group=None, tail_params=(), early_tail=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.owners = list(owners)
        self.tail = list(tail_arenas)
        # stand-alone parameters outside every arena (the fp32 log_A / post_norm.weight of the Qwen3.5 GDN layers, a fp32 vision
        # tower): their gradients travel as ONE coalesced buffer in finish_step
        self.tail_params = [p for p in tail_params if p.requires_grad]
        self._pending = []
        self._done = set()
        self.comm_stream = None
        self._beside = False  # collectives of this step are running beside the compute stream
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        for m in self.owners:
            object.__setattr__(m, "_grad_ready", self._on_ready)
        self.enabled = self.world > 1
        self._sync_on = True
        # (trigger owner, arena): the arena is complete -- except for sparse embedding rows that travel separately -- once the trigger's backward ran
        self.early_tail = early_tail
        self._events = {}
        self._split_now = False       # this step exchanges the early_tail arena in two parts (decided in begin_step)
        self._ctl_stream = None       # the per-step token-count check runs here (begin_step)
        self.bucket_blocks = max(1, int(os.environ.get("MI355_DDP_BUCKET_BLOCKS", "7")))      # owner buckets per group (one window per group)
        self.window_launches = max(0, int(os.environ.get("MI355_DDP_WINDOW_LAUNCHES", "2")))  # per-tile NT launches behind a group; 0 = no windows (persistent kernel always)
        self.window_auto = "MI355_DDP_WINDOW_LAUNCHES" not in os.environ  # sized from the measured all-reduce / launch times (begin_step)
        self.window_wait = os.environ.get("MI355_DDP_WINDOW_WAIT", "1") != "0"
        self._timings = []            # per window: [comm start, comm done, compute open, compute close or None, launches]
        self._tok_check = None        # this step's token-count agreement, not yet read on the host (begin_step -> _verify_tokens)
        self._group = []              # complete buckets waiting for their group to fill
        self._group_events = []       # completion events of the groups, reused every step

    # ---------------------------------------------------------------- loss weighting / accumulation
    def loss_weight(self, n_tokens):
        """Factor that turns this rank's per-token MEAN loss into its share of the global-batch mean under the AVG exchange:
        n_r * world / sum_r n_r.  ``n_tokens``: python int or 0-d / 1-element tensor (stays on its device).  1.0 when single-process."""
        if not self.enabled:
            return 1.0
        if torch.is_tensor(n_tokens):
            mine = n_tokens.detach().reshape(1).to(torch.float32)
        else:
            dev = self._device()
            mine = torch.tensor([float(n_tokens)], dtype=torch.float32, device=dev)
        total = mine.clone()
        dist.all_reduce(total, op=dist.ReduceOp.SUM, group=self.group)
        return (mine * self.world / total.clamp_min(1.0)).reshape(())

    @contextlib.contextmanager
    def no_sync(self):
        """Micro-steps of a gradient-accumulation window that must NOT exchange: buckets keep accumulating locally."""
        prev, self._sync_on = self._sync_on, False
        try:
            yield self
        finally:
            self._sync_on = prev

    def _device(self):
        for m in self.owners:
            for p in m.parameters():
                return p.device
        for p in self.tail_params:
            return p.device
        return torch.device("cpu")

    # ---------------------------------------------------------------- helpers
    def _arena(self, module):
        from .ops import arena_for

        return arena_for(module)

    def _reduce(self, arena):
        buf = arena.grad
        if buf.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=buf.device)
            ev = self._events.get(id(arena))  # one event per bucket, reused every step
            if ev is None:
                ev = self._events[id(arena)] = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(buf.device))
            with torch.cuda.stream(self.comm_stream):
                self.comm_stream.wait_event(ev)
                if self.backend == "nccl":
                    dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
                    buf.div_(self.world)
            buf.record_stream(self.comm_stream)
        else:
            dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group)
            buf.div_(self.world)

    def _window_behind(self, device, start=None):
        """The collectives just enqueued run beside the compute stream: its next ``window_launches`` persistent-sized NT GEMMs take the per-tile kernel, then it waits for
        their completion event and returns to the persistent kernel (module docstring).  Launches already queued are in front of the buckets' events, i.e. finished
        before the collectives start.  ``start``: event recorded on the communication stream in front of the group's collectives (timing)."""
        if self.comm_stream is None or self.window_launches == 0:
            return
        from . import kernels as K

        if start is not None and self.window_auto:
            done = torch.cuda.Event(enable_timing=True)  # read back a step or two later (_retune_window): an event of its own, not one of the reused ones
        else:
            n = K._WINDOW.stats["windows"] % 8
            while len(self._group_events) <= n:
                self._group_events.append(torch.cuda.Event())
            done = self._group_events[n]
        done.record(self.comm_stream)
        stream = torch.cuda.current_stream(device)  # the compute stream, captured now: the window's end acts on IT, whichever stream issues the closing launch
        rec = None
        if start is not None and self.window_auto:
            opened, closed = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            opened.record(stream)
            rec = [start, done, opened, closed, self.window_launches, False]
            self._timings.append(rec)

        def on_close():
            if rec is not None and not K._WINDOW.closed_early:  # (a window cut short by finish_step says nothing about what a launch takes)
                rec[3].record(stream)
                rec[5] = True
            if self.window_wait:
                stream.wait_event(done)

        K.open_gemm_window(self.window_launches, on_close, stream=stream)
        self._beside = True

    @staticmethod
    def window_for(allreduce_ms, launch_ms, lo=1, hi=16):
        """Windowed launches that cover an all-reduce of ``allreduce_ms`` when one windowed (per-tile) launch lasts ``launch_ms``."""
        import math

        if not (allreduce_ms > 0 and launch_ms > 0):
            return lo
        return max(lo, min(hi, math.ceil(allreduce_ms / launch_ms)))

    def _retune_window(self):
        """Read the windows whose four events have completed (no host wait) and size the next step's windows from them."""
        if not self.window_auto or not self._timings:
            return
        ar, per = [], []
        keep = []
        for rec in self._timings:
            start, done, opened, closed, launches, was_closed = rec
            if not was_closed:
                continue  # (closed by finish_step before its launches ran out: no launch time to read)
            if not (done.query() and closed.query()):
                keep.append(rec)
                continue
            ar.append(start.elapsed_time(done))
            per.append(opened.elapsed_time(closed) / max(launches, 1))
        self._timings = keep[-32:]  # (the host runs about a step ahead of the device: a step's windows are usually read at the begin_step after next)
        if ar and per:
            per.sort()
            self.window_launches = self.window_for(max(ar), per[len(per) // 2])

    def _flush_group(self):
        """Hand the waiting buckets to the communication stream, back to back, and open the compute stream's window behind them."""
        if not self._group:
            return
        device = None
        start = None
        for ar in self._group:
            if ar.grad.is_cuda and start is None and self.window_auto and self.window_launches:
                if self.comm_stream is None:
                    self.comm_stream = torch.cuda.Stream(device=ar.grad.device)
                start = torch.cuda.Event(enable_timing=True)
                self.comm_stream.wait_stream(torch.cuda.current_stream(ar.grad.device))  # (the first bucket's own event would order it too; this puts `start` behind the compute it waits for)
                start.record(self.comm_stream)
            self._reduce(ar)
            if ar.grad.is_cuda:
                device = ar.grad.device
        self._group.clear()
        if device is not None:
            self._window_behind(device, start)

    def _on_ready(self, module):
        if not self.enabled or not self._sync_on:
            return
        self._verify_tokens()  # (before the first collective whose shape depends on the split decision)
        if self._split_now and module is self.early_tail[0]:
            top = self.early_tail[1]
            if id(top) not in self._done and top.trainable():
                self._done.add(id(top))
                top.untouched_to_zero()
                self._reduce(top)  # the dense part of the tied bucket: at once, under the blocks still to come, with a window of its own
                if top.grad.is_cuda:
                    self._window_behind(top.grad.device)
        ar = self._arena(module)
        if id(ar) in self._done:
            return
        self._done.add(id(ar))
        ar.untouched_to_zero()
        self._group.append(ar)
        if len(self._group) >= self.bucket_blocks:
            self._flush_group()

    # ---------------------------------------------------------------- the embedding's sparse part of a split bucket
    def splits(self, arena):
        """True while this step exchanges ``arena`` in two parts: its dense part has been (or is being) all-reduced, embedding rows go through
        ``gather_embedding``."""
        return self.enabled and self._sync_on and self._split_now and arena is self.early_tail[1] and id(arena) in self._done

    def split_pays(self, tokens, row_bytes=None):
        """The byte rule: exchange the tied bucket in two parts iff the exposed all-gather of the token rows delivers fewer bytes to a rank
        than the dense ring all-reduce of the whole bucket would move."""
        if self.early_tail is None or not self.enabled:
            return False
        forced = os.environ.get("MI355_DDP_SPLIT_TIED", "auto")
        if forced in ("on", "off"):
            return forced == "on"
        top = self.early_tail[1]
        if row_bytes is None:
            table = next(p for p in top.params if p.dim() == 2)  # the embedding / head matrix [V, d]
            row_bytes = table.shape[1] * table.element_size()
        bucket_bytes = top.grad.numel() * top.grad.element_size() if top.grad is not None else sum(p.numel() * p.element_size() for p in top.params)
        w = self.world
        sparse = (w - 1) * int(tokens) * (row_bytes + 8)
        dense = 2.0 * (w - 1) / w * bucket_bytes
        return sparse < dense

    def _check_equal_tokens(self, tokens):
        """all_gather_into_tensor needs the same count on every rank.  One 16-byte MAX all-reduce of (T, -T), EVERY step that passes a token count (whatever this
        rank's byte rule says: the check itself must be collective).  With RCCL it runs on a control stream of its own and the host does NOT wait for it here: the
        result stays on the device until ``_verify_tokens`` reads it -- in front of the step's first bucket hand-off, i.e. before any collective whose shape depends
        on the split decision, and by then long complete -- so a step has no host synchronisation in ``begin_step`` (round 5 read it back at once)."""
        if self.backend == "nccl":
            dev = self._device()
            if self._ctl_stream is None:
                self._ctl_stream = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(self._ctl_stream):
                t = torch.tensor([int(tokens), -int(tokens)], dtype=torch.int64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
                host = torch.empty(2, dtype=torch.int64, pin_memory=True)
                host.copy_(t, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(self._ctl_stream)
            t.record_stream(self._ctl_stream)
            self._tok_check = (host, ev)
        else:
            t = torch.tensor([int(tokens), -int(tokens)], dtype=torch.int64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
            self._tok_check = (t, None)
            self._verify_tokens()

    def _verify_tokens(self):
        chk, self._tok_check = self._tok_check, None
        if chk is None:
            return
        host, ev = chk
        if ev is not None:
            ev.synchronize()  # the control stream's 16-byte collective of begin_step: finished long before the first bucket is complete
        hi, lo = host.tolist()
        if hi != -lo:
            raise RuntimeError(f"GradSync: ranks hold different embedding token counts ({-lo}..{hi}); the split tied-weight exchange all-gathers "
                               "equal shards -- pad the batches alike or call begin_step() without embedding_tokens (dense exchange)")

    def gather_embedding(self, ids, rows):
        """Every rank's token ids [T] and token-gradient rows [T, width], concatenated in rank order, and the factor 1 / world: what the
        deterministic embedding backward sums on every rank alike.  The dense all-reduce of the split bucket runs on the communication stream;
        the rows are added to that bucket afterwards, so the current stream waits for it here.  All ranks hold equally many tokens
        (``begin_step`` checked it)."""
        self._verify_tokens()
        flat = ids.reshape(-1).contiguous()
        rows = rows.contiguous()
        ids_all = torch.empty(self.world * flat.numel(), dtype=flat.dtype, device=flat.device)
        rows_all = torch.empty((self.world * rows.shape[0], rows.shape[1]), dtype=rows.dtype, device=rows.device)
        dist.all_gather_into_tensor(ids_all, flat, group=self.group)
        dist.all_gather_into_tensor(rows_all, rows, group=self.group)
        if self.comm_stream is not None:
            torch.cuda.current_stream(rows.device).wait_stream(self.comm_stream)
        return ids_all, rows_all, 1.0 / self.world

    # ---------------------------------------------------------------- step protocol
    def begin_step(self, embedding_tokens=None, embedding_row_bytes=None):
        """``embedding_tokens``: this rank's token count of the tied embedding's backward (``ids.numel()``); given it, the tied bucket is
        exchanged in two parts when ``split_pays`` says so.  Every rank must pass the same value (checked on first use)."""
        global _ACTIVE
        self._done.clear()
        self._group.clear()
        self._split_now = False
        self._retune_window()
        if embedding_tokens is not None and self.enabled and self._sync_on and self.early_tail is not None:
            self._check_equal_tokens(int(embedding_tokens))  # collective: every rank that passes a count takes part, whatever it then decides
            self._split_now = self.split_pays(embedding_tokens, embedding_row_bytes)
        _ACTIVE = self

    def finish_step(self):
        """Reduce buckets not yet sent, then order the compute stream after the communication stream."""
        if not self.enabled or not self._sync_on:
            return
        self._verify_tokens()
        self._flush_group()  # a partly filled group (the adapter's bucket, the blocks behind the last full group)
        for m in self.owners:  # anything whose hook never fired (e.g. unused in this step)
            ar = self._arena(m)
            if id(ar) not in self._done and ar.trainable():
                self._done.add(id(ar))
                ar.untouched_to_zero()
                self._reduce(ar)
        for ar in self.tail:
            if id(ar) not in self._done and ar.trainable():
                self._done.add(id(ar))
                ar.untouched_to_zero()
                self._reduce(ar)
        self._reduce_tail_params()
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)
        if self._beside:  # the streams have joined: whatever window is still open ends here
            from . import kernels as K

            K.close_gemm_window()
            self._beside = False
        global _ACTIVE
        _ACTIVE = None

    def _reduce_tail_params(self):
        ps = self.tail_params
        if not ps:
            return
        by_dtype = {}
        for p in ps:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
            by_dtype.setdefault(p.grad.dtype, []).append(p)
        for group_ps in by_dtype.values():
            flat = torch.cat([p.grad.reshape(-1) for p in group_ps])  # bucket assembly (communication plumbing)
            if flat.is_cuda:  # same hand-off as the arenas: the exchange and the copy back run on the communication stream
                if self.comm_stream is None:
                    self.comm_stream = torch.cuda.Stream(device=flat.device)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(flat.device))
                ctx = torch.cuda.stream(self.comm_stream)
                self.comm_stream.wait_event(ev)
            else:
                ctx = contextlib.nullcontext()
            with ctx:
                if self.backend == "nccl":
                    dist.all_reduce(flat, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group)
                    flat.div_(self.world)
                off = 0
                for p in group_ps:
                    n = p.grad.numel()
                    p.grad.copy_(flat[off : off + n].view_as(p.grad))
                    off += n
            if flat.is_cuda:
                flat.record_stream(self.comm_stream)
                for p in group_ps:
                    p.grad.record_stream(self.comm_stream)

    def broadcast_parameters(self, modules, src=0):
        """Make every replica start from rank ``src``'s weights (one broadcast per arena / parameter)."""
        if not self.enabled:
            return
        seen = set()
        for mod in modules:
            for p in mod.parameters():
                if p.data_ptr() in seen:
                    continue
                seen.add(p.data_ptr())
                dist.broadcast(p.data, src=src, group=self.group)


def sync_for_vlm(vlm_model, adapter):
    """GradSync wired for the early-fusion step: Qwen3 blocks (last -> first), adapter, then the embedding/head arena."""
    vlm_model._build_arenas()
    owners = list(reversed(list(vlm_model.trf_blocks))) + [adapter]
    # the head / embedding arena is complete, but for the embedding's token rows, when the last block's backward has run
    return GradSync(owners, tail_arenas=[vlm_model._top_arena], early_tail=(owners[0], vlm_model._top_arena))


def sync_for_qwen35(vlm):
    """GradSync for ``Qwen3_5VLM`` / ``Qwen3_5TextModel`` (BASELINE config 5): the text blocks' bf16 arenas fire last -> first during
    backward, the embedding / head arena and every parameter outside an arena (fp32 GDN parameters, the fp32 vision tower) follow in
    ``finish_step``."""
    lm = getattr(vlm, "language_model", vlm)
    lm._build_arenas()
    in_arena = set()
    for ar in lm.arenas():
        in_arena.update(id(p) for p in ar.params)
    tail_params = [p for p in vlm.parameters() if id(p) not in in_arena]
    return GradSync(list(reversed(list(lm.trf_blocks))), tail_arenas=[lm._top_arena], tail_params=tail_params)
