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
single-process global-batch gradient when every rank has the same number of target tokens (the synthetic all-ones mask).
For ragged masks use ``token_weighted=True``: gradients are pre-scaled by this rank's token count and divided by the
all-reduced total.
"""

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world_size, local_rank).  world_size 1 -> no process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
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

    def __init__(self, owners, tail_arenas=(), group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.owners = list(owners)
        self.tail = list(tail_arenas)
        self._pending = []
        self._done = set()
        self.comm_stream = None
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        for m in self.owners:
            object.__setattr__(m, "_grad_ready", self._on_ready)
        self.enabled = self.world > 1

    # ---------------------------------------------------------------- helpers
    def _arena(self, module):
        from .ops import arena_for

        return arena_for(module)

    def _reduce(self, arena):
        buf = arena.grad
        if buf.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=buf.device)
            ev = torch.cuda.Event()
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

    def _on_ready(self, module):
        if not self.enabled:
            return
        ar = self._arena(module)
        if id(ar) in self._done:
            return
        self._done.add(id(ar))
        ar.untouched_to_zero()
        self._reduce(ar)

    # ---------------------------------------------------------------- step protocol
    def begin_step(self):
        self._done.clear()

    def finish_step(self):
        """Reduce buckets not yet sent, then order the compute stream after the communication stream."""
        if not self.enabled:
            return
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
        if self.comm_stream is not None:
            torch.cuda.current_stream().wait_stream(self.comm_stream)

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
    return GradSync(owners, tail_arenas=[vlm_model._top_arena])
