"""ViT -> LLM adapter and the ViT train/eval loop -- API of ``llm_quest/multimodal/vision_transformer/vit_engine.py``."""

import torch

from llm_quest_amd import _lib as L
from llm_quest_amd import kernels as K
from llm_quest_amd import ops, rng

BF16, F32 = torch.bfloat16, torch.float32


class _AdapterFn(torch.autograd.Function):
    """simple: y = x W^T (+b);  ffn: y = gelu(x W0^T (+b0)) W3^T (+b3).  bf16, same rounding points as the reference
    (Linear output rounded to bf16, then nn.GELU on the bf16 tensor)."""

    @staticmethod
    def forward(ctx, x, mod, keep, *params):
        arena = ops.arena_for(mod)
        shp = x.shape
        x2 = x.reshape(-1, shp[-1]).contiguous()
        lins = mod._linears()
        bias = [None if l.bias is None else K.cast(l.bias.detach(), F32) for l in lins]
        if len(lins) == 1:
            y = K.gemm(L.GEMM_NT, x2, lins[0].weight, bias=bias[0])
            saved = (x2,)
        else:
            y1, a = K.gemm_gelu_dual(x2, lins[0].weight, bias=bias[0])  # Linear + GELU in one launch
            drop = None
            if mod.training and mod._dropout_p > 0:  # nn.Dropout between GELU and the second Linear (reference vit_engine.py:51)
                drop = (mod._dropout_p,) + rng.draw()
                a = K.dropout(a, *drop)
            y = K.gemm(L.GEMM_NT, a, lins[1].weight, bias=bias[1])
            saved = (x2, y1, a, drop)
        ctx.mod, ctx.saved, ctx.shp, ctx.need_dx = mod, saved if keep else None, shp, x.requires_grad
        return y.view(*shp[:-1], y.shape[-1])

    @staticmethod
    def backward(ctx, dy):
        mod = ctx.mod
        arena = ops.arena_for(mod)
        lins = mod._linears()
        dy2 = dy.reshape(-1, dy.shape[-1]).contiguous()

        def bias_grad(lin, g):
            if lin.bias is not None and lin.bias.requires_grad:
                view, acc = arena.grad_target(lin.bias)
                K.add_f32_to_bf16(K.colsum(g), view if acc else None, view)

        dx = None
        if len(lins) == 1:
            (x2,) = ctx.saved
            ops._wgrad(arena, lins[0].weight, None, dy2, x2)
            bias_grad(lins[0], dy2)
            if ctx.need_dx:
                dx = K.dgrad(dy2, lins[0].weight)
        else:
            x2, y1, a, drop = ctx.saved
            ops._wgrad(arena, lins[1].weight, None, dy2, a)
            bias_grad(lins[1], dy2)
            if drop is None:
                dy1 = K.gemm_dgrad_gelu_bwd(dy2, lins[1].weight, y1)  # GELU backward in the dgrad epilogue
            else:  # the same mask on the gradient between the dgrad and the GELU backward
                dy1 = K.gelu_bwd(y1, K.dropout(K.dgrad(dy2, lins[1].weight), *drop))
            ops._wgrad(arena, lins[0].weight, None, dy1, x2)
            bias_grad(lins[0], dy1)
            if ctx.need_dx:
                dx = K.dgrad(dy1, lins[0].weight)
        ctx.saved = None
        hook = getattr(mod, "_grad_ready", None)
        if hook is not None:
            hook(mod)
        return (None if dx is None else dx.view(ctx.shp), None, None) + (None,) * len(mod._param_list)


class ViTAdapter(torch.nn.Module):
    """Connector from ViT hidden size to the LLM embedding size (reference: vit_engine.py:9-59); keys ``adapter.weight``
    or ``adapter.{0,3}.weight``."""

    def __init__(self, vit_d_out, llm_d_in, adapter_type="simple", hidden_size_factor=4, bias=False, dropout=0.0, dtype=torch.float32):
        super().__init__()
        if adapter_type == "simple":
            self.adapter = torch.nn.Linear(vit_d_out, llm_d_in, bias=bias, dtype=dtype)
        elif adapter_type == "ffn":
            self.adapter = torch.nn.Sequential(
                torch.nn.Linear(vit_d_out, vit_d_out * hidden_size_factor, bias=bias, dtype=dtype),
                torch.nn.GELU(),
                torch.nn.Dropout(dropout) if dropout > 0.0 else torch.nn.Identity(),
                torch.nn.Linear(vit_d_out * hidden_size_factor, llm_d_in, bias=bias, dtype=dtype),
            )
        else:
            raise ValueError(f"Invalid adapter type: {adapter_type}")
        self._dropout_p = dropout if adapter_type == "ffn" else 0.0

    def _linears(self):
        return [self.adapter] if isinstance(self.adapter, torch.nn.Linear) else [self.adapter[0], self.adapter[3]]

    def forward(self, x):
        L.require_gpu(x)
        w = self._linears()[0].weight
        if w.dtype != BF16:
            raise TypeError("the HIP adapter computes in bf16: construct ViTAdapter(..., dtype=torch.bfloat16) (as the VLM step needs to feed Qwen3)")
        if x.dtype != BF16:
            x = K.cast(x.contiguous(), BF16)  # the reference casts the fp32 ViT states before a bf16 adapter too
        if not hasattr(self, "_param_list"):
            object.__setattr__(self, "_param_list", list(self.parameters()))
        return _AdapterFn.apply(x, self, torch.is_grad_enabled(), *self._param_list)


def vit_training_eval_loop(train_loader, val_loader, model, optimizer, num_epoch, lr_scheduler, eval_freq, eval_iter, device, use_amp=True):
    """ViT train/eval loop with LR schedule, clip(1.0) and per-epoch accuracy (reference: vit_engine.py:62-147).

    The HIP ViT always computes with bf16 MFMA operands over fp32 master weights -- exactly what ``use_amp=True`` asks
    of autocast upstream -- so ``use_amp`` is accepted for API parity and does not change the kernels.
    """
    from llm_quest_amd.engine import _cross_entropy, clip_grad_norm_

    step = 0
    train_losses, val_losses, train_accus, val_accus = [], [], [], []
    for epoch in range(1, num_epoch + 1):
        model.train()
        for input_batch, targets in train_loader:
            input_batch, targets = input_batch.to(device), targets.to(device)
            logits = model(input_batch)
            loss = _cross_entropy(logits, targets)
            optimizer.zero_grad()
            loss.backward()
            clip_grad_norm_(model.parameters(), max_norm=1)
            lr_scheduler.step(step)
            optimizer.step()
            step += 1
            if step == 1 or step % eval_freq == 0:
                tr, va = ViT.evaluate(train_loader, val_loader, model, eval_iter, device)
                train_losses.append(tr)
                val_losses.append(va)
                print(f"Epoch: {epoch}, Step: {step}", f"Train loss: {tr:.5f}, Val loss: {va:.5f}", f"lr: {lr_scheduler.current_lr:.1e}")
    # accuracy once after the last epoch, as upstream (its block sits outside the epoch loop)
    train_accus.append(ViT.accuracy_loader(train_loader, model, device))
    val_accus.append(ViT.accuracy_loader(val_loader, model, device))
    print(f"training accu epoch {num_epoch}: {train_accus[-1]*100:.4f}%", f"validation accu epoch {num_epoch}: {val_accus[-1]*100:.4f}%")
    return train_losses, val_losses, train_accus, val_accus


class ViT:
    """Evaluation helpers with the reference's static-method API (vit_engine.py:150-265)."""

    @staticmethod
    def accuracy_loader(data_loader, model, device):
        model.eval()
        correct = total = 0
        with torch.no_grad():
            for X, y in data_loader:
                pred = model(X.to(device)).argmax(dim=-1)
                correct += int((pred == y.to(device)).sum())
                total += len(pred)
        return correct / total

    @staticmethod
    def evaluate(train_loader, val_loader, model, eval_iter, device):
        model.eval()
        with torch.no_grad():
            out = (ViT.calc_loss_loader(train_loader, model, device, num_batches=eval_iter), ViT.calc_loss_loader(val_loader, model, device, num_batches=eval_iter))
        model.train()
        return out

    @staticmethod
    def calc_loss_loader(dataloader, model, device, num_batches=None):
        if len(dataloader) == 0:
            return float("NaN")
        num_batches = len(dataloader) if num_batches is None else min(num_batches, len(dataloader))
        total = 0.0
        for i, (X, y) in enumerate(dataloader):
            if i >= num_batches:
                break
            total += ViT._calc_loss_batch(X, y, model, device).item()
        return total / num_batches

    @staticmethod
    def _calc_loss_batch(X, y, model, device):
        """CE of one classification batch (reference vit_engine.py:246-265); the HIP CE kernel for device logits."""
        from llm_quest_amd.engine import _cross_entropy

        return _cross_entropy(model(X.to(device)), y.to(device))
