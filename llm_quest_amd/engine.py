"""Train / eval engine -- API of ``llm_quest/engine.py`` (same function names, arguments and return values).

Host-side control flow only; everything per-batch on the device goes through the HIP kernels.  Models that are plain
PyTorch modules on the CPU (BASELINE config 1: GPT-2 plumbing) run through the same loops unchanged.
"""

import math
import time

import torch

from . import ops


def _cross_entropy(logits2d, targets):
    """CE with ignore_index=-100: HIP kernel for bf16 device logits, torch for host tensors (CPU plumbing config)."""
    if logits2d.is_cuda and logits2d.dtype == torch.bfloat16:
        t = targets.reshape(-1).contiguous()
        lg = logits2d if (logits2d.stride(-1) == 1 and logits2d.stride(0) % 8 == 0) else _pad_rows(logits2d)
        return ops.CrossEntropyFn.apply(lg, t).to(logits2d.dtype)  # the reference returns the loss in the logits dtype
    if logits2d.is_cuda and logits2d.dtype == torch.float32 and not logits2d.requires_grad:
        from . import kernels as K  # eval-mode fp32 logits (e.g. the ViT classifier head): evaluate through the bf16 CE kernel

        return _cross_entropy(K.cast(logits2d.contiguous(), torch.bfloat16), targets).float()
    return torch.nn.functional.cross_entropy(logits2d, targets)


class _PadRowsFn(torch.autograd.Function):
    """Copy [rows, V] into a buffer whose row pitch is a multiple of 8 elements (16-byte rows for the CE kernel)."""

    @staticmethod
    def forward(ctx, x):
        from . import kernels as K

        rows, v = x.shape
        pitch = (v + 7) // 8 * 8
        buf = torch.empty((rows, pitch), dtype=x.dtype, device=x.device)
        K.copy2d(x if x.stride(-1) == 1 else x.contiguous(), buf[:, :v])
        ctx.v = v
        return buf[:, :v]

    @staticmethod
    def backward(ctx, g):
        from . import kernels as K

        out = torch.empty((g.shape[0], ctx.v), dtype=g.dtype, device=g.device)
        K.copy2d(g if g.stride(-1) == 1 else g.contiguous(), out)
        return out


def _pad_rows(x):
    return _PadRowsFn.apply(x)


def global_loss(logits, y, model=None, classification=False):
    """CE (+ summed MoE auxiliary losses when present) (reference: engine.py:50-72)."""
    loss = _cross_entropy(logits, y) if classification else _cross_entropy(logits.flatten(0, 1), y.flatten())
    aux = 0.0
    for module in model.modules():  # model=None raises, as upstream does
        ffn = getattr(module, "ffn", None)
        if ffn is not None and hasattr(ffn, "moe_loss"):
            aux = aux + ffn.moe_loss
    return loss + aux


def _calc_loss_batch(X, y, model, device, attn_mask=None, classification=False):
    X, y = X.to(device), y.to(device)
    if attn_mask is not None:
        attn_mask = attn_mask.to(device)
    if classification:
        return _cross_entropy(model(X, last_token_only=True, attn_mask=attn_mask), y)
    return _cross_entropy(model(X, attn_mask=attn_mask).flatten(0, 1), y.flatten())


def calc_loss_loader(dataloader, model, device, num_batches=None, classification=False):
    if len(dataloader) == 0:
        return float("NaN")
    num_batches = len(dataloader) if num_batches is None else min(num_batches, len(dataloader))
    total = 0.0
    for i, batch in enumerate(dataloader):
        if i >= num_batches:
            break
        if len(batch) == 3:
            X, y, m = batch
            total += _calc_loss_batch(X, y, model, device, m, classification).item()
        else:
            X, y = batch
            total += _calc_loss_batch(X, y, model, device, classification=classification).item()
    return total / num_batches


class LearningRateScheduler:
    """Linear warm-up then optional cosine decay; ``step(step)`` sets the lr OF that step (reference: engine.py:114-202)."""

    def __init__(self, optimizer, total_steps, init_lr, peak_lr, warmup_steps=0, min_lr=None, decay=None):
        if warmup_steps > 0 and init_lr >= peak_lr:
            raise ValueError(f"Warmup enabled (warmup_steps={warmup_steps}) but init_lr ({init_lr:.2e}) >= peak_lr ({peak_lr:.2e}).")
        if min_lr is not None and min_lr >= peak_lr:
            raise ValueError(f"min_lr ({min_lr:.2e}) >= peak_lr ({peak_lr:.2e}).")
        if decay is not None and min_lr is None:
            raise ValueError(f"decay='{decay}' was set but min_lr=None.")
        if decay is None and min_lr is not None:
            raise ValueError(f"min_lr ({min_lr:.2e}) was set but decay=None.")
        self.optimizer, self.total_steps, self.peak_lr = optimizer, total_steps, peak_lr
        self.warmup_steps = max(warmup_steps, 0)
        self.init_lr = init_lr if self.warmup_steps > 0 else peak_lr
        self.current_lr = self.init_lr
        self.warmup_range = self.peak_lr - self.init_lr
        self.lr_step = self.warmup_range / self.warmup_steps if self.warmup_steps > 0 else 0
        self.min_lr = peak_lr if min_lr is None else min_lr
        self.decay = decay if min_lr is not None else None
        for group in self.optimizer.param_groups:
            group["lr"] = self.current_lr

    def _get_cosine_decay_lr(self, step):
        span = self.total_steps - self.warmup_steps
        return self.min_lr + (self.peak_lr - self.min_lr) * 0.5 * (1 + math.cos(math.pi * (step - self.warmup_steps) / span))

    def step(self, step):
        if step < self.warmup_steps:
            self.current_lr = self.init_lr + self.lr_step * step
            if step + 1 == self.warmup_steps:
                print(f"Warmup finished at step {step+1}. Peak LR reached: {self.peak_lr:.1e}")
        else:
            self.current_lr = self._get_cosine_decay_lr(step) if self.decay == "cosine" else self.peak_lr
        for group in self.optimizer.param_groups:
            if not group.get("custom_lr", False):
                group["lr"] = self.current_lr


def _unpack(batch, device):
    if len(batch) == 3:
        X, y, m = batch
        return X.to(device), y.to(device), m.to(device)
    X, y = batch
    return X.to(device), y.to(device), None


def _amp(device, use_amp):
    """bf16 models on the HIP path already compute in explicit bf16; autocast only matters for fp32 host models."""
    dev_type = device.type if isinstance(device, torch.device) else str(device)
    return torch.autocast(dev_type, dtype=torch.bfloat16, enabled=use_amp and dev_type == "cpu")


def evaluate(train_loader, val_loader, model, eval_iter, device, classification=False):
    model.eval()
    with torch.no_grad():
        tr = calc_loss_loader(train_loader, model, device, num_batches=eval_iter, classification=classification)
        va = calc_loss_loader(val_loader, model, device, num_batches=eval_iter, classification=classification)
    model.train()
    return tr, va


def training_eval_loop_simple(train_loader, val_loader, model, optimizer, num_epoch, eval_freq, eval_iter, device):
    step, train_losses, val_losses = 0, [], []
    for epoch in range(1, num_epoch + 1):
        model.train()
        for batch in train_loader:
            step += 1
            X, y, m = _unpack(batch, device)
            loss = global_loss(model(X, attn_mask=m), y, model=model)
            loss.backward()
            optimizer.step()
            optimizer.zero_grad()
            if step % eval_freq == 0:
                tr, va = evaluate(train_loader, val_loader, model, eval_iter, device)
                train_losses.append(tr)
                val_losses.append(va)
                print(f"Epoch: {epoch}, Step: {step}", f"Train loss: {tr:.5f}, Val loss: {va:.5f}")


def training_eval_loop_simple_timing(train_loader, val_loader, model, optimizer, num_epoch, eval_freq, eval_iter, device):
    """tok/s benchmark loop: HIP-event timed intervals, first interval discarded (reference: engine.py:270-374)."""
    step = last_tokens = total_tokens = 0
    train_losses, val_losses, track_tokens = [], [], []
    cum_tokens = cum_time = 0.0
    on_gpu = (device.type if isinstance(device, torch.device) else str(device)) == "cuda"
    if on_gpu:
        t_start, t_end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t_start.record()
    else:
        t0 = time.time()
    for epoch in range(1, num_epoch + 1):
        model.train()
        for X, y in train_loader:
            step += 1
            X, y = X.to(device), y.to(device)
            logits = model(X)
            optimizer.zero_grad()
            loss = global_loss(logits, y, model=model)
            loss.backward()
            optimizer.step()
            total_tokens += X.numel()
            if step % eval_freq == 0:
                if on_gpu:
                    t_end.record()
                    torch.cuda.synchronize()
                    elapsed = t_start.elapsed_time(t_end) / 1000
                    t_start.record()
                else:
                    elapsed, t0 = time.time() - t0, time.time()
                interval = total_tokens - last_tokens
                last_tokens = total_tokens
                tps = interval / elapsed if elapsed > 0 else 0
                if step > 1:
                    cum_tokens += interval
                    cum_time += elapsed
                avg = cum_tokens / cum_time if cum_time > 0 else 0
                tr, va = evaluate(train_loader, val_loader, model, eval_iter, device)
                train_losses.append(tr)
                val_losses.append(va)
                print(f"Epoch: {epoch}, Step: {step}", f"Train loss: {tr:.5f}, Val loss: {va:.5f}", f"Step tok/sec: {round(tps)}, Avg tok/sec: {round(avg)}")
        if on_gpu:
            print(f"\nAllocated memory: {torch.cuda.memory_allocated() / 1024**3:.4f} GB")
            print(f"Reserved memory: {torch.cuda.memory_reserved() / 1024**3:.4f} GB\n")
    return train_losses, val_losses, track_tokens


def clip_grad_norm_(parameters, max_norm):
    """torch.nn.utils.clip_grad_norm_ semantics; on device tensors the global norm and the rescale are HIP kernels and
    nothing synchronises with the host."""
    params = [p for p in parameters if p.grad is not None]
    if not params:
        return None
    if not params[0].grad.is_cuda:
        return torch.nn.utils.clip_grad_norm_(params, max_norm)
    from . import kernels as K

    acc = torch.zeros(1, dtype=torch.float32, device=params[0].grad.device)
    seen = set()
    grads = []
    for p in params:
        g = p.grad
        if g.data_ptr() in seen:
            continue
        seen.add(g.data_ptr())
        grads.append(g if g.is_contiguous() else g.contiguous())
        K.sumsq_into(grads[-1], acc)
    for g in grads:
        K.clip_scale_(g, acc, max_norm)
    return acc.sqrt()


def training_eval_loop(train_loader, val_loader, model, optimizer, num_epoch, lr_scheduler, eval_freq, eval_iter, device, accumulation_steps=1, use_amp=True):
    """Grad accumulation (ragged last window), clip 1.0, scheduler step before optimizer step, eval at step 1 and every
    eval_freq (reference: engine.py:377-470)."""
    step, train_losses, val_losses = 0, [], []
    for epoch in range(1, num_epoch + 1):
        model.train()
        for i, batch in enumerate(train_loader):
            is_last = i == len(train_loader) - 1
            acc_pos = (i + 1) % accumulation_steps
            X, y, m = _unpack(batch, device)
            with _amp(device, use_amp):
                loss = global_loss(model(X, attn_mask=m), y, model=model)
                loss = loss / acc_pos if (is_last and acc_pos != 0) else loss / accumulation_steps
            loss.backward()
            if acc_pos == 0 or is_last:
                clip_grad_norm_(model.parameters(), max_norm=1)
                lr_scheduler.step(step)
                optimizer.step()
                optimizer.zero_grad()
                step += 1
                if step == 1 or step % eval_freq == 0:
                    tr, va = evaluate(train_loader, val_loader, model, eval_iter, device)
                    train_losses.append(tr)
                    val_losses.append(va)
                    print(f"Epoch: {epoch}, Step: {step}  | ", f"Train loss: {tr:.5f}  Val loss: {va:.5f}  | ", f"lr: {lr_scheduler.current_lr:.1e}")
    return train_losses, val_losses


def profile_training_eval_loop(train_loader, val_loader, model, optimizer, num_epoch, warmup_percent, init_lr, peak_lr, min_lr, eval_freq,
                               eval_iter, device, use_amp=True, profile_dir="./torch_profile_logs", wait=1, warmup=1, active=3, repeat=1,
                               record_shapes=True, profile_memory=True):
    """A short training run under ``torch.profiler`` (reference: engine.py:499-640): inline linear warm-up / cosine schedule
    over ``len(train_loader) * num_epoch`` steps, one optimizer step per batch with clip 1.0, eval every ``eval_freq`` steps,
    and an early stop once ``wait + warmup + active`` steps were traced.  Kernel-level numbers for the HIP path come from
    rocprofv3; this entry gives the host-side timeline the reference's users expect."""
    import os

    from torch.profiler import ProfilerActivity, profile

    os.makedirs(profile_dir, exist_ok=True)
    dev_type = device.type if isinstance(device, torch.device) else str(device).split(":")[0]
    acts = [ProfilerActivity.CPU] + ([ProfilerActivity.CUDA] if dev_type == "cuda" else [])
    total_steps = len(train_loader) * num_epoch
    warmup_steps = int(warmup_percent * total_steps)
    lr_step = (peak_lr - init_lr) / warmup_steps if (warmup_percent and warmup_steps) else 0.0
    budget = wait + warmup + active
    train_losses, val_losses = [], []
    step, traced, done = -1, 0, False
    with profile(activities=acts, schedule=torch.profiler.schedule(wait=wait, warmup=warmup, active=active, repeat=repeat),
                 on_trace_ready=torch.profiler.tensorboard_trace_handler(profile_dir), record_shapes=record_shapes,
                 profile_memory=profile_memory, with_stack=True) as prof:
        for epoch in range(1, num_epoch + 1):
            model.train()
            for X, y in train_loader:
                step += 1
                if step < warmup_steps:
                    lr = init_lr + step * lr_step
                else:
                    lr = min_lr + (peak_lr - min_lr) * 0.5 * (1 + math.cos(math.pi * (step - warmup_steps) / (total_steps - warmup_steps)))
                for group in optimizer.param_groups:
                    if not group.get("custom_lr", False):
                        group["lr"] = lr
                X, y = X.to(device), y.to(device)
                with _amp(device, use_amp):
                    loss = global_loss(model(X), y, model=model)
                optimizer.zero_grad()
                loss.backward()
                clip_grad_norm_(model.parameters(), max_norm=1)
                optimizer.step()
                if step % eval_freq == 0:
                    tr, va = evaluate(train_loader, val_loader, model, eval_iter, device)
                    train_losses.append(tr)
                    val_losses.append(va)
                    print(f"Epoch: {epoch}, Step: {step}", f"Train loss: {tr:.5f}, Val loss: {va:.5f}")
                prof.step()
                if traced >= budget:
                    done = True
                    break
                traced += 1
            if done:
                break
    print(f"Profiling complete. Logs saved to {profile_dir}")
    return train_losses, val_losses
