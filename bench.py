"""Headline benchmark: img+tokens/sec of the early-fusion VLM forward+backward step (BASELINE config 4).

    python bench.py --gpus N --steps K --warmup W

N > 1 runs one rank per GPU over RCCL.  Under ``torch.distributed.run`` (WORLD_SIZE set) this process IS a rank; started
plainly with ``--gpus N`` it starts the N ranks itself -- a child ``python -m torch.distributed.run --nproc-per-node N bench.py ...``
spawned before this process has touched the GPU -- and relays rank 0's JSON line and the child's exit status.

A step = vision tower (frozen ViT-B/16, eval) -> ffn adapter -> early-fusion concat -> Qwen3-0.6B decoder on the fused
709-token sequence -> tied LM head + cross entropy on the 512 text positions -> full backward (adapter + LLM), plus the
RCCL gradient all-reduce when N > 1.  The optimizer step is NOT part of the metric; gradients are
dropped between steps (zero_grad(set_to_none=True)).  Synthetic seeded inputs, random-init weights, inputs resident in
HBM before the timed region.  One JSON line on rank 0.
"""

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_VISION, N_TEXT, VOCAB = 197, 512, 151_936
UNITS_PER_SAMPLE = 1 + N_TEXT  # 1 image + 512 text tokens
ALGO_FLOP_PER_SAMPLE = 2.566e12  # 35.13 G (frozen ViT fwd) + 3 x (2.17 G adapter + 841.5 G LLM)
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)
CPU_SAMPLES = 8  # size of the bounded CPU-baseline sample (~10-15 s on the box's 16-core quota)


def build_models(device, seed=123):
    from llm_quest_amd.config import VIT_BASE_CONFIG, qwen3_config_creator
    from llm_quest_amd.multimodal.vision_transformer.vit_engine import ViTAdapter
    from llm_quest_amd.multimodal.vision_transformer.vit_model import ViTModel
    from llm_quest_amd.qwen.qwen3.qwen3_model import Qwen3Model

    torch.manual_seed(seed)
    vit_cfg = dict(VIT_BASE_CONFIG, drop_rate=0.0)
    llm_cfg = dict(qwen3_config_creator("0.6B"), context_length=1024)
    with torch.device(device):
        vit = ViTModel(vit_cfg)
        llm = Qwen3Model(llm_cfg)
        ad = ViTAdapter(768, 1024, adapter_type="ffn", dtype=torch.bfloat16)
    vit.eval()
    for p in vit.parameters():
        p.requires_grad = False
    llm.train()
    ad.train()
    return vit, vit_cfg, ad, llm, llm_cfg


def synthetic_batch(batch, device, seed, ragged=False):
    """SURVEY 8d, config 4 inputs: randn images, uniform token ids, all-ones text mask; ``ragged``: real lengths ~ U[256, 512], right-padded."""
    g = torch.Generator().manual_seed(seed)
    img = torch.randn(batch, 3, 224, 224, generator=g)
    ids = torch.randint(0, VOCAB, (batch, N_TEXT), generator=g)
    mask = torch.ones(batch, N_TEXT, dtype=torch.bool)
    if ragged:
        lengths = torch.randint(256, N_TEXT + 1, (batch,), generator=g)
        mask = torch.arange(N_TEXT).unsqueeze(0) < lengths.unsqueeze(1)
    return img.to(device), ids.to(device), mask.to(device)


def usable_cores():
    """CPU threads this process may really use: affinity mask capped by the cgroup CPU quota (GPU boxes share a host)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def _cpu_baseline_worker(state_path, seed, threads, q):
    """Child process: the CPU oracle (oracle/models.py) on one full-size sample, forward + backward."""
    torch.set_num_threads(threads)
    from llm_quest_amd.config import VIT_BASE_CONFIG, qwen3_config_creator
    from oracle import models

    st = torch.load(state_path)
    vit_cfg = dict(VIT_BASE_CONFIG, drop_rate=0.0)
    llm_cfg = dict(qwen3_config_creator("0.6B"), context_length=1024)
    ad_sd = {k: v.requires_grad_(True) for k, v in st["ad"].items()}
    llm_sd = {k: v.requires_grad_(True) for k, v in st["llm"].items()}
    llm_sd["out_head.weight"] = llm_sd["emb_dict.weight"]
    img, ids, mask = synthetic_batch(CPU_SAMPLES, "cpu", seed)

    def step(i, t, m):
        loss, _, _ = models.vlm_forward_loss(st["vit"], vit_cfg, ad_sd, llm_sd, llm_cfg, i, t, m)
        loss.backward()
        return float(loss.detach())

    step(img[:1], ids[:1, :16], mask[:1, :16])  # warm the thread pool / allocator on a short sequence
    t0 = time.perf_counter()
    loss = step(img, ids, mask)
    q.put((time.perf_counter() - t0, loss))


def cpu_baseline(vit, ad, llm, seed, budget_s=240):
    """The CPU oracle (proved equal to the reference on fixtures) timed on this host's cores on a BOUNDED sample: one
    full-size sample (1 image + 512 tokens, S=709), forward+backward, same weights.  Runs in a child process with a hard
    time budget so the default bench always finishes in minutes; reported, never the optimisation target."""
    import tempfile

    import torch.multiprocessing as mp

    cores = min(usable_cores(), 32)
    state = {
        "vit": {k: v.detach().cpu() for k, v in vit.state_dict().items()},
        "ad": {k: v.detach().cpu().clone() for k, v in ad.state_dict().items()},
        "llm": {k: v.detach().cpu().clone() for k, v in llm.state_dict().items() if k not in ("mask", "cos", "sin", "out_head.weight")},
    }
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "state.pt")
        torch.save(state, path)
        del state
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        proc = ctx.Process(target=_cpu_baseline_worker, args=(path, seed, cores, q))
        proc.start()
        proc.join(budget_s)
        if proc.is_alive():
            proc.kill()
            proc.join()
            return {"value": None, "unit": "img+tok/s", "cores": cores, "kind": "port",
                    "sample": f"{CPU_SAMPLES} full-size samples fwd+bwd did not finish within the {budget_s} s budget on {cores} threads"}
        dt, loss = q.get(timeout=10)
    return {
        "value": round(CPU_SAMPLES * UNITS_PER_SAMPLE / dt, 3), "unit": "img+tok/s", "cores": cores, "kind": "port",
        "sample": f"{CPU_SAMPLES} samples (each 1 img + 512 tok, S=709) full-size fwd+bwd in {dt:.1f} s on {cores} threads, torch {torch.__version__} CPU, oracle loss {loss:.4f}",
    }


def _profile_json(name):
    """A committed counter file from profiles/ -- only if it was measured on the kernel sources this process runs
    (llm_quest_amd/fingerprint.py); a stale file yields None, never a number."""
    from llm_quest_amd.fingerprint import kernel_sources_sha

    path = os.path.join(ROOT, "profiles", name)
    try:
        with open(path) as f:
            d = json.load(f)
    except (OSError, ValueError):
        return None, f"{name} not found"
    if d.get("kernel_sources_sha") != kernel_sources_sha():
        return None, f"{name} was measured on other kernel sources (its fingerprint {d.get('kernel_sources_sha')} != {kernel_sources_sha()}): re-collect it"
    return d, None


def pmc_traffic():
    """TCC counters of the dominant kernel, collected with rocprofv3 --pmc in separate passes and committed (PMC cannot be read
    from inside the timed process)."""
    from llm_quest_amd.fingerprint import EVIDENCE_ROUND

    d, why = _profile_json(f"{EVIDENCE_ROUND}_pmc_tcc_gemm.json")
    return (d["kernels"] if d else None), why


def pmc_step_traffic():
    """Whole-step TCC counters (profiles/<round>_pmc_tcc_step.json), collected with rocprofv3 --pmc over this script."""
    from llm_quest_amd.fingerprint import EVIDENCE_ROUND

    return _profile_json(f"{EVIDENCE_ROUND}_pmc_tcc_step.json")


def dominant_kernel_rate(batch, device):
    """HIP-event timing of the step's dominant kernel class (gemm_bf16_kernel) on its largest shapes, on the stream the
    kernels are launched on (torch's current stream)."""
    from llm_quest_amd import _lib as L
    from llm_quest_amd import kernels as K

    M = batch * (N_VISION + N_TEXT)
    x = torch.randn(M, 1024, device=device).to(torch.bfloat16)
    w = torch.randn(6144, 1024, device=device).to(torch.bfloat16)
    dy = torch.randn(M, 6144, device=device).to(torch.bfloat16)
    out = {}
    for name, fn, flops in (
        ("NT gate-up fwd", lambda: K.gemm(L.GEMM_NT, x, w), 2.0 * M * 6144 * 1024),
        ("NT gate-up dgrad on W^T (the step's form, transpose included)", lambda: K.dgrad(dy, w), 2.0 * M * 6144 * 1024),
        ("NN gate-up dgrad (form not used by the step)", lambda: K.gemm(L.GEMM_NN, dy, w), 2.0 * M * 6144 * 1024),
        ("TN gate-up wgrad", lambda: K.gemm(L.GEMM_TN, dy, x), 2.0 * M * 6144 * 1024),
    ):
        for _ in range(3):
            fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10):
            fn()
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / 10
        out[name] = {"ms": round(ms, 4), "tflops": round(flops / ms / 1e9, 1)}
    return out


def launch_ranks(n):
    """``python bench.py --gpus N`` without a torchrun environment: run the N ranks as a child ``torch.distributed.run`` job (a fresh
    process tree; nothing in THIS process has initialised the GPU yet) and pass its output and exit status through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=160, help="per-GPU micro-batch (samples).  160 x 709 tokens keep 165 GiB of the 288 GB of HBM live; same-box pairs: 64 -> 34.2 / 34.1 %%, "
                    "128 -> +1.4 %%, 160 -> 35.0 / 34.9 %%, 192 -> level with 160 (202 GiB).  More rounds of tiles per launch amortise every launch's fill and tail.  (92 -- every "
                    "launch a whole number of rounds at batch ~64 -- measured the same as 64: the tail tiles of a partial round run faster, the chip is power-limited)")
    ap.add_argument("--cpu-baseline", choices=["auto", "off"], default="auto")
    ap.add_argument("--ragged", action="store_true", help="text lengths ~ U[256, 512] (padding mask active in attention and loss) instead of all-ones masks")
    ap.add_argument("--optimizer", choices=["on", "off"], default="on", help="also time the same steps with clip + AdamW (reported beside the fwd+bwd metric)")
    ap.add_argument("--vision-ahead", choices=["on", "off"], default="on", help="frozen ViT of the next step's batch on a second stream (vlm_engine.VisionAhead), as the training loop runs it")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    # MI355_DDP_REHEARSAL=1 (one-GPU boxes only): the N ranks share the visible GPUs round-robin and exchange over gloo -- every line of the N > 1 path
    # (rank start-up, broadcast, bucketed exchange on the communication stream, split tied-weight bucket, barriers, max-over-ranks timing, the one
    # JSON line) runs except RCCL itself.  The line it prints carries "rehearsal": true and is no measurement.
    rehearsal = os.environ.get("MI355_DDP_REHEARSAL") == "1"
    if rehearsal and args.gpus > torch.cuda.device_count() and args.batch * args.gpus > 192 * max(torch.cuda.device_count(), 1):
        raise SystemExit("rehearsal: the ranks share the visible GPU(s) -- pass a small --batch (e.g. 8), the default would not fit")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        n_dev = torch.cuda.device_count()  # counts devices without creating a HIP context
        if n_dev < args.gpus and not rehearsal:
            raise SystemExit(f"--gpus {args.gpus} but only {n_dev} GPU(s) are visible")
        raise SystemExit(launch_ranks(args.gpus))

    from llm_quest_amd import _lib, ddp
    from llm_quest_amd.multimodal.vlm_engine import VisionAhead, vlm_step_loss

    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:  # before any rendezvous: a mismatched launch must fail at once, not wait for peers
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={os.environ.get('WORLD_SIZE', '1')}: start one rank per GPU (python bench.py --gpus N does it itself)")
    rank, world, local = ddp.init_from_env("gloo" if rehearsal else None)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if rehearsal:
        local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    _lib.load()

    vit, vit_cfg, ad, llm, llm_cfg = build_models(device)
    sync = ddp.sync_for_vlm(llm, ad)
    sync.broadcast_parameters([llm, ad, vit])
    img, ids, mask = synthetic_batch(args.batch, device, seed=123 + rank, ragged=args.ragged)

    n_targets = mask.sum()  # this rank's target tokens: ragged shards weight their mean loss by it (ddp.GradSync.loss_weight)
    # The frozen tower runs one batch ahead on a second stream, exactly as vlm_training_loop_simple runs it: every step takes the hidden
    # states submitted during the previous step and submits the next batch's (here the same synthetic images) -- one ViT forward per
    # step, inside the timed region, concurrent with the decoder instead of in front of it.
    ahead = VisionAhead(vit) if args.vision_ahead == "on" else None
    if ahead is not None:
        ahead.submit(img)

    def vision():
        if ahead is None:
            return None
        h = ahead.take(img)
        ahead.submit(img)
        return h

    def step():
        loss = vlm_step_loss(vit, llm, ad, img, ids, mask, hf_vit_model=False, vit_hidden=vision())
        sync.begin_step(embedding_tokens=ids.numel())  # the tied head / embedding bucket: dense or in two parts, by bytes (ddp.GradSync.split_pays)
        (loss * sync.loss_weight(n_targets) if (args.ragged and world > 1) else loss).backward()
        sync.finish_step()
        llm.zero_grad(set_to_none=True)
        ad.zero_grad(set_to_none=True)
        return loss

    def fence():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        loss = step()
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        loss = step()
    ev1.record()
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    loss_gpu = float(loss.detach())

    # The metric is forward + backward (BASELINE.json; SURVEY 8d: "optimizer step excluded (state it)").  So that nothing of a real
    # training step is left unmeasured, the same K steps are timed again WITH the optimizer: global-norm clip 1.0 + AdamW on the
    # parameter arenas (llm_quest_amd/optim.py::ArenaAdamW), reported beside `value`, never instead of it.
    train_step = None
    if args.optimizer == "on":
        from llm_quest_amd.optim import ArenaAdamW

        opt = ArenaAdamW(list(llm.parameters()) + list(ad.parameters()), lr=1e-5, weight_decay=0.01, max_grad_norm=1.0)
        opt.attach(llm, ad)

        def full_step():
            loss_ = vlm_step_loss(vit, llm, ad, img, ids, mask, hf_vit_model=False, vit_hidden=vision())
            sync.begin_step(embedding_tokens=ids.numel())
            (loss_ * sync.loss_weight(n_targets) if (args.ragged and world > 1) else loss_).backward()
            sync.finish_step()
            opt.step()
            opt.zero_grad(set_to_none=True)
            return loss_

        for _ in range(max(args.warmup, 1)):
            full_step()
        fence()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            full_step()
        fence()
        el2 = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([el2], dtype=torch.float64, device=device)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            el2 = float(t)
        train_step = {"what": "the same steps including the optimizer: global-norm clip 1.0 + AdamW (ArenaAdamW, one fused launch per parameter arena)",
                      "ms_per_step": round(el2 / args.steps * 1e3, 3), "value": round(world * args.batch * UNITS_PER_SAMPLE * args.steps / el2, 1), "unit": "img+tok/s"}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.batch * UNITS_PER_SAMPLE * args.steps / elapsed
        achieved = ALGO_FLOP_PER_SAMPLE * args.batch / (elapsed / args.steps) / 1e12  # per GPU
        line = {
            "metric": "img+tokens/sec fwd+bwd, ViT-B+Qwen3-0.6B VLM, 224px+512tok",
            "value": round(value, 1), "unit": "img+tok/s", "n_gpus": world, "rccl_ranks": world if (world > 1 and not rehearsal) else 0, "steps": args.steps, "warmup": args.warmup,
            **({"rehearsal": True, "rehearsal_note": "ranks share the visible GPU(s) and exchange over gloo: a run of the N > 1 code path, not a measurement"} if rehearsal else {}),
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[3]: VLM early fusion, ViT-B/16 (frozen, fwd) + ffn adapter 768->3072->1024 + Qwen3-0.6B, "
                            "224x224 image + 512 text tokens (S=709), fwd+loss+bwd, no optimizer step",
                "per_gpu_batch": args.batch, "global_batch": args.batch * world, "parallelism": f"dp{world}", "text_mask": "ragged U[256,512]" if args.ragged else "all ones",
                "units_per_sample": UNITS_PER_SAMPLE,
                "vision_tower": "frozen ViT forward of the NEXT step's batch on a second HIP stream, one per timed step (vlm_engine.VisionAhead)" if ahead is not None else "in front of the decoder, same stream",
            },
            "roofline": {
                "bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                "traffic_note": "not collected",
                "basis": "algorithmic 2.566 TFLOP/sample (SURVEY 8d) x per-GPU batch / step time; device-side (HIP events) "
                         f"{dev_ms / args.steps:.3f} ms/step",
            },
            "loss": round(loss_gpu, 5), "peak_memory_gib": round(torch.cuda.max_memory_allocated(device) / 2**30, 1),
        }
        if train_step is not None:
            line["with_optimizer_step"] = train_step
        if world == 1:
            line["roofline"]["dominant_kernel"] = {"name": "gemm_bf16_kernel", "hip_event_timing": dominant_kernel_rate(args.batch, device)}
            pmc, why_gemm = pmc_traffic()
            step_pmc, why_step = pmc_step_traffic()
            pmc_batch = step_pmc.get("per_gpu_batch") if step_pmc else None
            if step_pmc is not None and pmc_batch != args.batch:  # the PMC passes were taken at another batch's shapes
                line["roofline"]["traffic_note"] = f"the committed counter passes were taken at per-GPU batch {pmc_batch}"
            elif step_pmc is None:
                line["roofline"]["traffic_note"] = "traffic withheld: " + why_step
            else:
                line["roofline"]["traffic"] = step_pmc["per_step"]["total_bytes"]
                line["roofline"]["traffic_note"] = (
                    "memory-side bytes PER STEP (like `achieved`): 2 x FETCH_SIZE (gfx950 correction) + WRITE_SIZE summed over every kernel of one step, separate "
                    "rocprofv3 --pmc passes over this bench (profiles/<round>_pmc_tcc_step.json, per-kernel table inside; fingerprint of the kernel sources checked); "
                    "Infinity-Cache hits are counted in FETCH_SIZE")
            if pmc is not None and pmc_batch == args.batch:
                line["roofline"]["dominant_kernel"]["pmc_bytes_per_launch"] = {f: pmc[f]["hbm_bytes"] for f in pmc}
                line["roofline"]["dominant_kernel"]["over_algorithmic"] = {f: pmc[f]["over_algorithmic"] for f in pmc}
            elif pmc is None:
                line["roofline"]["dominant_kernel"]["pmc_note"] = "withheld: " + why_gemm
        if world == 1 and args.cpu_baseline == "auto":
            try:
                line["cpu_baseline"] = cpu_baseline(vit, ad, llm, seed=123)
            except Exception as exc:  # the baseline is a reported number, never the measured path
                line["cpu_baseline"] = {"value": None, "unit": "img+tok/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {exc!r}"}
        print(json.dumps(line), flush=True)
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
