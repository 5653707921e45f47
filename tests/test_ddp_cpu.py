"""world_size-2 data-parallel path on the CPU (gloo): the bucket protocol of llm_quest_amd.ddp.GradSync.

The same code drives RCCL on the GPUs; here the arenas hold CPU tensors, so no HIP kernel is involved -- what is checked
is the protocol: hooks fire per bucket, every bucket is averaged exactly once per step, tail buckets and never-fired
owners are flushed in finish_step, parameters are broadcast from rank 0, and the result equals the single-process
global-batch gradient.
"""

import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _Owner(torch.nn.Module):
    def __init__(self, seed):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.a = torch.nn.Parameter(torch.randn(16, 8, generator=g))
        self.b = torch.nn.Parameter(torch.randn(8, generator=g))


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from llm_quest_amd.ddp import GradSync, init_from_env
    from llm_quest_amd.ops import arena_for

    r, w, _ = init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    owners = [_Owner(100 + rank + i) for i in range(3)]  # different init per rank on purpose
    tail_owner = _Owner(50 + rank)
    loose = [torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(2, 3))]  # parameters outside every arena
    sync = GradSync(owners, tail_arenas=[arena_for(tail_owner)], tail_params=loose)
    sync.broadcast_parameters(owners + [tail_owner])
    ref0 = _Owner(100)  # rank 0's first owner
    assert torch.equal(owners[0].a, ref0.a)

    def fake_backward(step):
        sync.begin_step()
        for i, m in enumerate(owners[:2]):  # owner 2 never fires -> must be flushed by finish_step
            ar = arena_for(m)
            for p in m.parameters():
                view, acc = ar.grad_target(p)
                val = torch.full_like(p, float(rank + 1 + i + step))
                view.copy_(view + val if acc else val)
            m._grad_ready(m)
        ar = arena_for(tail_owner)
        view, acc = ar.grad_target(tail_owner.a)
        view.copy_(torch.full_like(view, 10.0 * (rank + 1)))
        loose[0].grad = torch.full((5,), float(rank + 1))  # loose[1] gets no gradient on purpose: reduced as zeros
        sync.finish_step()

    fake_backward(0)
    expect = lambda i, step: sum(rk + 1 + i + step for rk in range(world)) / world
    ok = True
    for i, m in enumerate(owners[:2]):
        ok &= bool(torch.allclose(m.a.grad, torch.full_like(m.a, expect(i, 0))))
        ok &= bool(torch.allclose(m.b.grad, torch.full_like(m.b, expect(i, 0))))
    ok &= owners[2].a.grad is not None and float(owners[2].a.grad.abs().sum()) == 0.0
    ok &= bool(torch.allclose(tail_owner.a.grad, torch.full_like(tail_owner.a, 10.0 * sum(range(1, world + 1)) / world)))
    ok &= float(tail_owner.b.grad.abs().sum()) == 0.0
    ok &= bool(torch.allclose(loose[0].grad, torch.full((5,), sum(range(1, world + 1)) / world)))
    ok &= loose[1].grad is not None and float(loose[1].grad.abs().sum()) == 0.0
    # second step after zero_grad(set_to_none=True): buckets are overwritten, not accumulated
    for m in owners + [tail_owner]:
        for p in m.parameters():
            p.grad = None
    for p in loose:
        p.grad = None
    fake_backward(1)
    ok &= bool(torch.allclose(owners[0].a.grad, torch.full_like(owners[0].a, expect(0, 1))))
    out.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gradsync_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def test_gradsync_single_process_is_a_noop():
    from llm_quest_amd.ddp import GradSync

    m = _Owner(1)
    s = GradSync([m])
    assert s.enabled is False
    s.begin_step()
    m._grad_ready(m)
    s.finish_step()


def _ragged_worker(rank, world, port, out):
    """Token-weighted exchange + gradient accumulation against the single-process global batch, on real autograd graphs."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from llm_quest_amd.ddp import GradSync, init_from_env
    from llm_quest_amd.ops import arena_for

    init_from_env(backend="gloo")
    torch.manual_seed(7)
    lin = torch.nn.Linear(6, 4)  # the same weights on every rank (seeded)
    g = torch.Generator().manual_seed(11)
    X = torch.randn(2 * world, 5, 6, generator=g)  # global batch: 2 samples per rank, 5 positions each
    Y = torch.randint(0, 4, (2 * world, 5), generator=g)
    lengths = torch.tensor([5, 2, 1, 4][: 2 * world])  # ragged: rank 0 has 7 target tokens, rank 1 has 5
    M = torch.arange(5).unsqueeze(0) < lengths.unsqueeze(1)

    def mean_loss(x, y, m):
        return torch.nn.functional.cross_entropy(lin(x).flatten(0, 1), y.masked_fill(~m, -100).flatten(), ignore_index=-100)

    # reference: one process, the whole batch
    want_w, want_b = torch.autograd.grad(mean_loss(X, Y, M), [lin.weight, lin.bias])
    # data parallel: this rank's shard, gradients written into the arena by autograd's .grad views
    ar = arena_for(lin)
    sync = GradSync([lin])

    def shard_backward(weighted):
        lin.zero_grad(set_to_none=True)
        ar.grad.zero_()
        sl = slice(2 * rank, 2 * rank + 2)
        loss = mean_loss(X[sl], Y[sl], M[sl])
        w = sync.loss_weight(M[sl].sum()) if weighted else 1.0
        sync.begin_step()
        grads = torch.autograd.grad(loss * w, list(lin.parameters()))
        for p, g_ in zip(lin.parameters(), grads):  # what the wgrad kernels do on the GPU: the gradient is written into the arena
            view, _ = ar.grad_target(p)
            view.copy_(g_)
        lin._grad_ready(lin)
        sync.finish_step()

    shard_backward(weighted=True)
    ok = bool(torch.allclose(lin.weight.grad, want_w, atol=1e-6)) and bool(torch.allclose(lin.bias.grad, want_b, atol=1e-6))
    shard_backward(weighted=False)  # the plain average of per-rank means is NOT the global-batch gradient on ragged shards
    ok &= not bool(torch.allclose(lin.weight.grad, want_w, atol=1e-4))
    # accumulation window of two micro-steps: nothing is exchanged inside no_sync, the sums are exchanged once
    lin.zero_grad(set_to_none=True)
    ar.grad.zero_()
    with sync.no_sync():
        sync.begin_step()
        view, _ = ar.grad_target(lin.bias)
        view.copy_(torch.full_like(view, float(rank + 1)))
        lin._grad_ready(lin)
        sync.finish_step()
    ok &= bool(torch.equal(lin.bias.grad, torch.full_like(lin.bias, float(rank + 1))))  # still local
    sync.begin_step()
    view, acc = ar.grad_target(lin.bias)
    ok &= acc is True
    view.add_(10.0)
    lin._grad_ready(lin)
    sync.finish_step()
    ok &= bool(torch.allclose(lin.bias.grad, torch.full_like(lin.bias, sum(r + 1 for r in range(world)) / world + 10.0)))
    out.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_token_weighted_exchange_and_no_sync_world2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_ragged_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert sorted(results) == [(0, True), (1, True)]


def _split_worker(rank, world, port, out):
    """The tied head / embedding bucket in two parts (ddp.GradSync.early_tail): the dense part is exchanged when the trigger owner's backward has
    run, the embedding's token rows are all-gathered and summed by every rank alike."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    from llm_quest_amd import ddp
    from llm_quest_amd.ops import arena_for

    ddp.init_from_env(backend="gloo")
    V, W, T = 12, 4, 6
    owners = [_Owner(7 + i) for i in range(2)]

    class _Top(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Parameter(torch.zeros(V, W))
            self.norm = torch.nn.Parameter(torch.zeros(W))

    top = _Top()
    sync = ddp.GradSync(owners, tail_arenas=[arena_for(top)], early_tail=(owners[0], arena_for(top)))
    g = torch.Generator().manual_seed(40 + rank)
    head_grad = torch.randn(V, W, generator=g)          # this rank's LM-head weight gradient
    norm_grad = torch.randn(W, generator=g)
    ids = torch.randint(0, V, (T,), generator=g)        # this rank's tokens (ids repeat across and inside ranks)
    rows = torch.randn(T, W, generator=g)

    def step(exchange, tokens=T):
        for p in list(top.parameters()) + [q for m in owners for q in m.parameters()]:
            p.grad = None
        sync.begin_step(embedding_tokens=tokens)
        ar = arena_for(top)
        ar.grad_target(top.emb)[0].copy_(head_grad)     # first gradient of the backward
        ar.grad_target(top.norm)[0].copy_(norm_grad)
        assert not sync.splits(ar)
        for m in owners:                                 # blocks, last to first: owners[0] is the trigger
            for p in m.parameters():
                arena_for(m).grad_target(p)[0].fill_(float(rank + 1))
            m._grad_ready(m)
        # the embedding backward, last of all
        view, _ = ar.grad_target(top.emb)
        if sync.splits(ar):
            assert exchange
            ids_all, rows_all, scale = sync.gather_embedding(ids, rows)
            view.index_add_(0, ids_all, rows_all * scale)  # what mi355_embedding_bwd_sorted does on the device
        else:
            assert not exchange
            view.index_add_(0, ids, rows)
        sync.finish_step()
        return top.emb.grad.clone(), top.norm.grad.clone(), owners[0].a.grad.clone()

    emb, norm, blk = step(True)
    # expectation: the mean over ranks of (head gradient + scatter of the rank's rows)
    parts = []
    for rk in range(world):
        gg = torch.Generator().manual_seed(40 + rk)
        hg, ng = torch.randn(V, W, generator=gg), torch.randn(W, generator=gg)
        ii = torch.randint(0, V, (T,), generator=gg)
        rr = torch.randn(T, W, generator=gg)
        parts.append((hg.index_add(0, ii, rr), ng))
    want_emb = sum(p[0] for p in parts) / world
    want_norm = sum(p[1] for p in parts) / world
    ok = bool(torch.allclose(emb, want_emb, atol=1e-6)) and bool(torch.allclose(norm, want_norm, atol=1e-6))
    ok &= bool(torch.allclose(blk, torch.full_like(blk, sum(range(1, world + 1)) / world)))
    ok &= ddp.active() is None
    # the byte rule (ddp.GradSync.split_pays): the 224-byte bucket costs a 2-rank ring 224 bytes per rank, T = 6 rows of 16 + 8 bytes cost the
    # all-gather 144 -> split (above); 10 T rows would cost 1 440 -> the bucket stays whole and goes out dense in finish_step; no count -> whole
    ok &= sync.split_pays(T) and not sync.split_pays(10 * T)
    for hint in (10 * T, None):
        emb_d, norm_d, _ = step(False, tokens=hint)
        ok &= bool(torch.allclose(emb_d, want_emb, atol=1e-6)) and bool(torch.allclose(norm_d, want_norm, atol=1e-6))
    # unequal token counts across ranks are refused before any all-gather can hang
    try:
        sync.begin_step(embedding_tokens=T + rank)
        ok &= world == 1
    except RuntimeError as exc:
        ok &= "different embedding token counts" in str(exc)
    # accumulation window: nothing is exchanged, the bucket keeps this rank's own sums
    with sync.no_sync():
        emb_l, _, _ = step(False)
    ok &= bool(torch.allclose(emb_l, head_grad.index_add(0, ids, rows), atol=1e-6))
    out.put((rank, ok))


def test_tied_head_embedding_bucket_is_exchanged_in_two_parts_gloo():
    world = 2
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs) and res == {0: True, 1: True}


def _world8_worker(rank, world, port, out):
    """The headline step's hand-offs at EIGHT ranks (the 8-rank arithmetic had only ever run with two): 28 block buckets last -> first in groups of
    7, the tied head / embedding bucket in two parts (64 x 512 tokens: the byte rule says split at 8 ranks) and whole (160 x 512: dense), the adapter's bucket and the
    rest in finish_step -- every gradient the 8-rank mean."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    from llm_quest_amd import ddp
    from llm_quest_amd.ops import arena_for

    ddp.init_from_env(backend="gloo")
    V, W, T = 12, 4, 6
    blocks = [_Owner(7 + i) for i in range(28)]
    adapter = _Owner(99)

    class _Top(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.emb = torch.nn.Parameter(torch.zeros(V, W))
            self.norm = torch.nn.Parameter(torch.zeros(W))

    top = _Top()
    owners = blocks + [adapter]  # (already last -> first)
    sync = ddp.GradSync(owners, tail_arenas=[arena_for(top)], early_tail=(owners[0], arena_for(top)))
    flushed = []
    inner = sync._flush_group
    sync._flush_group = lambda: (flushed.append(len(sync._group)), inner())[1]
    g = torch.Generator().manual_seed(40 + rank)
    head_grad, norm_grad = torch.randn(V, W, generator=g), torch.randn(W, generator=g)
    ids, rows = torch.randint(0, V, (T,), generator=g), torch.randn(T, W, generator=g)

    def step(split):
        flushed.clear()
        for p in list(top.parameters()) + [q for m in owners for q in m.parameters()]:
            p.grad = None
        os.environ["MI355_DDP_SPLIT_TIED"] = "on" if split else "off"  # (the tiny bucket's own byte rule is checked in the world-2 test; the real sizes below)
        sync.begin_step(embedding_tokens=T)
        ar = arena_for(top)
        ar.grad_target(top.emb)[0].copy_(head_grad)
        ar.grad_target(top.norm)[0].copy_(norm_grad)
        for i, m in enumerate(owners):
            for p in m.parameters():
                arena_for(m).grad_target(p)[0].fill_(float(rank + 1 + i))
            m._grad_ready(m)
        view, _ = ar.grad_target(top.emb)
        if sync.splits(ar):
            ids_all, rows_all, scale = sync.gather_embedding(ids, rows)
            view.index_add_(0, ids_all, rows_all * scale)
        else:
            view.index_add_(0, ids, rows)
        sync.finish_step()
        return top.emb.grad.clone(), [m.a.grad.clone() for m in owners], list(flushed)

    parts = []
    for rk in range(world):
        gg = torch.Generator().manual_seed(40 + rk)
        hg, _ng = torch.randn(V, W, generator=gg), torch.randn(W, generator=gg)
        parts.append(hg.index_add(0, torch.randint(0, V, (T,), generator=gg), torch.randn(T, W, generator=gg)))
    want_emb = sum(parts) / world
    ok = True
    for split in (True, False):
        emb, grads, fl = step(split)
        ok &= bool(torch.allclose(emb, want_emb, atol=1e-6))
        for i, gr in enumerate(grads):
            ok &= bool(torch.allclose(gr, torch.full_like(gr, sum(rk + 1 + i for rk in range(world)) / world)))
        ok &= [n for n in fl if n] == [7, 7, 7, 7, 1]  # four groups of block buckets during the backward, the adapter's bucket in finish_step
    # the byte rule at the real sizes (no memory: meta tensors): 8 ranks, bucket = [151 936, 1 024] bf16 + the final norm
    import types

    real = types.SimpleNamespace(grad=None, params=[torch.empty(151936, 1024, dtype=torch.bfloat16, device="meta"), torch.empty(1024, dtype=torch.bfloat16, device="meta")])
    os.environ["MI355_DDP_SPLIT_TIED"] = "auto"
    probe = ddp.GradSync([_Owner(1)], early_tail=(None, real))
    ok &= probe.world == 8 and probe.split_pays(64 * 512) and not probe.split_pays(160 * 512)
    out.put((rank, ok))
    dist.barrier()
    dist.destroy_process_group()


def test_headline_step_hand_offs_at_eight_ranks_gloo():
    world = 8
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(60)
    assert all(p.exitcode == 0 for p in procs) and res == {r: True for r in range(world)}


def test_persistent_gemm_switch_restores_the_users_threshold(monkeypatch):
    """kernels.persistent_gemm(False) keeps the library off the persistent NT kernel (whose workgroups must all start together) while collectives run beside
    the compute stream; persistent_gemm(True) puts back what the user had set -- nothing, or their own threshold."""
    import os

    from llm_quest_amd import kernels as K

    var = "MI355_GEMM_PERSIST_MIN_TILES"
    monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(K, "_PERSIST_USER", None)
    K.persistent_gemm(False)
    assert os.environ[var] == "1000000000"
    K.persistent_gemm(True)
    assert var not in os.environ
    monkeypatch.setattr(K, "_PERSIST_USER", "64")
    K.persistent_gemm(False)
    assert os.environ[var] == "1000000000"
    K.persistent_gemm(True)
    assert os.environ[var] == "64"


def test_gemm_window_bounds_the_per_tile_stretch_behind_a_bucket_group(monkeypatch):
    """kernels.open_gemm_window(k, on_close): the next k persistent-sized NT launches stay off the persistent kernel, the launch after them first runs the
    hand-off (the compute stream's wait for the group's collectives) and re-enables it; small launches do not count; a window opened inside a window extends
    it and both hand-offs run; close_gemm_window ends it at once (GradSync.finish_step)."""
    import os

    from llm_quest_amd import kernels as K

    var = "MI355_GEMM_PERSIST_MIN_TILES"
    monkeypatch.delenv(var, raising=False)
    monkeypatch.setattr(K, "_PERSIST_USER", None)
    monkeypatch.setattr(K, "_WINDOW", K._GemmWindow())
    big, small = (113440, 4096), (2048, 1024)
    calls = []
    K._nt_tick(*big)
    assert K._WINDOW.stats == {"persistent_eligible": 1, "inside_window": 0, "windows": 0} and var not in os.environ
    K.open_gemm_window(2, lambda: calls.append("a"))
    assert os.environ[var] == "1000000000"
    K._nt_tick(*small)  # 32 tiles: never persistent, not counted
    K._nt_tick(*big)
    K.open_gemm_window(2, lambda: calls.append("b"))  # a second group while the window is open: two more launches, both hand-offs at the end
    K._nt_tick(*big)
    K._nt_tick(*big)
    assert calls == [] and os.environ[var] == "1000000000" and K._WINDOW.stats["inside_window"] == 3
    K._nt_tick(*big)  # the first launch behind the window: hand-offs first, then persistent again
    assert calls == ["a", "b"] and var not in os.environ
    assert K._WINDOW.stats == {"persistent_eligible": 2, "inside_window": 3, "windows": 1}
    K.open_gemm_window(5, lambda: calls.append("c"))
    K.close_gemm_window()
    assert calls == ["a", "b", "c"] and var not in os.environ and K._WINDOW.left == 0
    K.close_gemm_window()  # idempotent
    assert calls == ["a", "b", "c"]


def test_gemm_window_counts_only_persistent_eligible_launches_and_sizes_itself(monkeypatch):
    """a launch the library would NOT put on the persistent kernel (bias, fp32 output, a tile hint ...: ``eligible=False``) does not use the window up; the
    user's own MI355_GEMM_PERSIST_MIN_TILES is the threshold; GradSync.window_for sizes a window from the measured all-reduce and launch times."""
    from llm_quest_amd import ddp
    from llm_quest_amd import kernels as K

    monkeypatch.delenv("MI355_GEMM_PERSIST_MIN_TILES", raising=False)
    monkeypatch.setattr(K, "_PERSIST_USER", None)
    monkeypatch.setattr(K, "_WINDOW", K._GemmWindow())
    big = (113440, 4096)
    K.open_gemm_window(2, None)
    K._nt_tick(*big, eligible=False)
    K._nt_tick(*big, eligible=False)
    assert K._WINDOW.left == 2 and K._WINDOW.stats["inside_window"] == 0
    K._nt_tick(*big)
    assert K._WINDOW.left == 1
    K.close_gemm_window()
    monkeypatch.setattr(K, "_PERSIST_USER", "100000")  # the user's threshold: 7 104 tiles are below it
    K.open_gemm_window(1, None)
    K._nt_tick(*big)
    assert K._WINDOW.left == 1
    K.close_gemm_window()
    wf = ddp.GradSync.window_for
    assert wf(1.3, 1.1) == 2 and wf(5.0, 1.1) == 5 and wf(0.2, 1.1) == 1 and wf(100.0, 1.0) == 16 and wf(0.0, 1.0) == 1 and wf(1.0, 0.0) == 1


def test_block_buckets_leave_in_groups(monkeypatch):
    """GradSync collects complete block buckets and hands them over MI355_DDP_BUCKET_BLOCKS at a time; finish_step flushes the rest."""
    from llm_quest_amd import ddp

    class FakeArena:
        def __init__(self):
            self.grad = torch.zeros(4)

        def untouched_to_zero(self):
            pass

        def trainable(self):
            return True

    owners = [torch.nn.Linear(2, 2) for _ in range(5)]
    arenas = {id(m): FakeArena() for m in owners}
    monkeypatch.setenv("MI355_DDP_BUCKET_BLOCKS", "2")
    sync = ddp.GradSync(owners)
    sync.enabled, sync.world = True, 2
    sent = []
    monkeypatch.setattr(sync, "_arena", lambda m: arenas[id(m)])
    monkeypatch.setattr(sync, "_reduce", lambda ar: sent.append(ar))
    monkeypatch.setattr(sync, "_reduce_tail_params", lambda: None)
    sync.begin_step()
    for i, m in enumerate(owners):
        m._grad_ready(m)
        assert len(sent) == 2 * ((i + 1) // 2), (i, len(sent))
    sync.finish_step()
    assert sent == [arenas[id(m)] for m in owners]
