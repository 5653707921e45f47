"""Early-fusion VLM train / eval step -- API of ``llm_quest/multimodal/vlm_engine.py``.

The reference wires ViT -> adapter -> GPT-2 (vlm_engine.py:44-164); BASELINE config 4 composes the same step with
Qwen3-0.6B as the language model.  Both work here:
  * an LLM exposing ``forward_hidden`` / ``lm_loss`` (llm_quest_amd Qwen3Model) takes the FAST path: fused embeddings
    are assembled by strided HIP copies, the decoder runs on the fused sequence with causal | ~key_mask, and the tied LM
    head + cross entropy run only on the rows that feed the loss (positions n_vision-1 .. S-2);
  * any other model (the GPT-2 plumbing model on the CPU) takes the generic path, line for line the reference's math.
"""

import os

import torch
import torch.nn.functional as F


def get_embeddings(text_input, model):
    """token + learned positional embeddings, GPT-2 only (reference: vlm_engine.py:5-20)."""
    tok = model.emb_dict(text_input)
    pos = model.pos_emb_dict(torch.arange(tok.shape[1], dtype=torch.long, device=tok.device))
    return tok + pos


def vlm_loss(logits, labels, text_attention_mask, num_vision_tokens):
    """CE of logits[:, n_vision-1:-1] vs labels with padded positions set to -100 (reference: vlm_engine.py:23-41)."""
    shifted = logits[:, num_vision_tokens - 1 : -1, :]
    tgt = labels.masked_fill(text_attention_mask == 0, -100)
    if shifted.is_cuda and shifted.dtype == torch.bfloat16:
        from llm_quest_amd import ops

        return ops.CrossEntropyFn.apply(shifted.flatten(0, 1), tgt.flatten().contiguous()).to(logits.dtype)
    return F.cross_entropy(shifted.flatten(0, 1), tgt.flatten(), ignore_index=-100)


def _is_native_llm(m):
    return hasattr(m, "forward_hidden") and hasattr(m, "lm_loss")


def _vision_states(vit_model, images, hf_vit_model):
    if hf_vit_model:
        return vit_model(images).last_hidden_state
    return vit_model(images, output_hidden_states=True)


def fuse_embeddings(vision_embeddings, text_embeddings):
    """cat([vision, text], dim=1) as two strided device copies (bit-exact) (reference: vlm_engine.py:111-114)."""
    from llm_quest_amd import kernels as K

    B, nv, d = vision_embeddings.shape
    nt = text_embeddings.shape[1]
    fused = torch.empty((B, nv + nt, d), dtype=text_embeddings.dtype, device=text_embeddings.device)
    K.copy2d(vision_embeddings.reshape(B, nv * d), fused.view(B, (nv + nt) * d)[:, : nv * d])
    K.copy2d(text_embeddings.reshape(B, nt * d), fused.view(B, (nv + nt) * d)[:, nv * d :])
    return fused


class _FuseFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vis, txt):
        ctx.nv = vis.shape[1]
        return fuse_embeddings(vis.contiguous(), txt.contiguous())

    @staticmethod
    def backward(ctx, g):
        from llm_quest_amd import kernels as K

        B, S, d = g.shape
        nv = ctx.nv
        g = g.contiguous()
        gv = torch.empty((B, nv, d), dtype=g.dtype, device=g.device)
        gt = torch.empty((B, S - nv, d), dtype=g.dtype, device=g.device)
        K.copy2d(g.view(B, S * d)[:, : nv * d], gv.view(B, nv * d))
        K.copy2d(g.view(B, S * d)[:, nv * d :], gt.view(B, (S - nv) * d))
        return gv, gt


KEEP_LOSS_ROWS = os.environ.get("MI355_KEEP_LOSS_ROWS", "1") != "0"  # A/B knob: 0 = every block on the whole sequence, rows gathered behind the final norm


class _RowsFn(torch.autograd.Function):
    """hidden[:, lo:hi, :] -> contiguous rows (and the scatter back in backward), as strided device copies."""

    @staticmethod
    def forward(ctx, h, lo, hi):
        from llm_quest_amd import kernels as K

        B, S, d = h.shape
        ctx.meta = (B, S, d, lo, hi)
        out = torch.empty((B, hi - lo, d), dtype=h.dtype, device=h.device)
        K.copy2d(h.contiguous().view(B, S * d)[:, lo * d : hi * d], out.view(B, (hi - lo) * d))
        return out

    @staticmethod
    def backward(ctx, g):
        from llm_quest_amd import kernels as K

        B, S, d, lo, hi = ctx.meta
        full = torch.zeros((B, S, d), dtype=g.dtype, device=g.device)
        K.copy2d(g.contiguous().view(B, (hi - lo) * d), full.view(B, S * d)[:, lo * d : hi * d])
        return full, None, None


class VisionAhead:
    """The FROZEN vision tower of the next batch on a second HIP stream, under the current batch's LLM forward / backward.

    The tower is frozen in this path (``requires_grad=False``, ``eval()``; reference vlm_engine.py:80-83), so its output for batch
    i+1 depends on nothing step i computes: ``submit(images)`` enqueues its forward on a side stream, ``take(images)`` hands the
    hidden states to the main stream (event hand-off, one tensor crosses).  The tower's launches are small (M = 197 tokens per image;
    most of its GEMM grids cover half the chip) and leave CUs idle when they run alone; next to the decoder's kernels they fill the
    tails of those grids instead.  Same work per step, same results (the kernels are deterministic); only WHEN the tower runs changes.
    """

    def __init__(self, vit_model, hf_vit_model=False):
        self.vit, self.hf = vit_model, hf_vit_model
        self.stream = None
        self.queue = []  # [(images, hidden, done event)], oldest first

    def submit(self, images):
        if any(p.requires_grad for p in self.vit.parameters()):
            raise RuntimeError("VisionAhead: the vision tower must be frozen (its forward runs outside autograd, one batch early)")
        dev = images.device
        if self.stream is None:
            self.stream = torch.cuda.Stream(device=dev)
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(dev))  # whatever produced `images` on the main stream
        with torch.cuda.stream(self.stream), torch.no_grad():
            self.stream.wait_event(ready)
            hidden = _vision_states(self.vit, images, self.hf)
            done = torch.cuda.Event()
            done.record(self.stream)
        images.record_stream(self.stream)
        self.queue.append((images, hidden, done))

    def take(self, images):
        """Hidden states of ``images``: the oldest submission if it was made for this very tensor, else computed now."""
        if self.queue and self.queue[0][0] is images:
            _, hidden, done = self.queue.pop(0)
            main = torch.cuda.current_stream(images.device)
            main.wait_event(done)
            hidden.record_stream(main)
            return hidden
        with torch.no_grad():
            return _vision_states(self.vit, images, self.hf)


def vlm_step_loss(vit_model, vlm_model, adapter, images, input_ids, text_attention_mask, hf_vit_model=False, vit_hidden=None):
    """Forward of one early-fusion step -> scalar loss (fp32 on the native path).  Used by the train and eval loops.
    ``vit_hidden``: the frozen tower's hidden states of ``images`` when they were computed ahead (``VisionAhead``)."""
    if vit_hidden is None:
        with torch.no_grad():
            vit_hidden = _vision_states(vit_model, images, hf_vit_model)
    vision_emb = adapter(vit_hidden)
    nv = vision_emb.shape[1]
    B = images.shape[0]
    if _is_native_llm(vlm_model):
        text_emb = vlm_model.emb_dict(input_ids)
        fused = _FuseFn.apply(vision_emb.to(text_emb.dtype), text_emb)
        mask = torch.cat([torch.ones(B, nv, dtype=torch.bool, device=fused.device), text_attention_mask.to(torch.bool)], dim=1)
        T = input_ids.shape[1]
        if KEEP_LOSS_ROWS:  # the last block's FFN half and the final norm run on the rows the loss reads (the last vision token predicts the first text token)
            rows = vlm_model.forward_hidden(fused, attn_mask=mask, input_embedded=True, keep_rows=(nv - 1, nv - 1 + T))
        else:
            hidden = vlm_model.forward_hidden(fused, attn_mask=mask, input_embedded=True)
            rows = _RowsFn.apply(hidden, nv - 1, nv - 1 + T)
        targets = input_ids.masked_fill(text_attention_mask == 0, -100)
        return vlm_model.lm_loss(rows.reshape(B * T, -1), targets.reshape(-1))
    text_emb = get_embeddings(input_ids, vlm_model)
    fused = torch.cat([vision_emb, text_emb], dim=1)
    mask = torch.cat([torch.ones(B, nv, dtype=torch.bool, device=fused.device), text_attention_mask], dim=1)
    logits = vlm_model(fused, attn_mask=mask, input_embedded=True)
    return vlm_loss(logits, input_ids, text_attention_mask, nv)


def vlm_training_loop_simple(vit_model, vlm_model, adapter, train_loader, optimizer, num_epochs, device, hf_vit_model=True,
                             val_loader=None, eval_freq=None, eval_iter=None, grad_sync=None):
    """Frozen ViT, trainable adapter + LLM; loss -> backward -> clip(1.0) -> step (reference: vlm_engine.py:44-164).

    ``grad_sync`` (optional, llm_quest_amd.ddp.GradSync) all-reduces the gradient buckets over RCCL, overlapped with
    backward, before the clip; absent -> single process, exactly the reference's step.
    """
    from llm_quest_amd.engine import clip_grad_norm_

    vit_model.eval()
    for p in vit_model.parameters():
        p.requires_grad = False
    vlm_model.train()
    adapter.train()
    vit_model.to(device)
    vlm_model.to(device)
    adapter.to(device)
    on_gpu = torch.device(device).type == "cuda"
    ahead = VisionAhead(vit_model, hf_vit_model) if on_gpu else None

    def batches(loader):
        """(batch on the device, its successor on the device or None): one batch of lookahead for the frozen tower."""
        moved = lambda b: None if b is None else {k: b[k].to(device) for k in ("image", "input_ids", "attention_mask")}
        it = iter(loader)
        cur = moved(next(it, None))
        while cur is not None:
            nxt = moved(next(it, None))
            yield cur, nxt
            cur = nxt

    for epoch in range(1, num_epochs + 1):
        total_loss = torch.zeros((), dtype=torch.float32, device=device)
        for step, (batch, nxt) in enumerate(batches(train_loader)):
            images, input_ids, mask = batch["image"], batch["input_ids"], batch["attention_mask"]
            vit_hidden = None
            if ahead is not None:
                vit_hidden = ahead.take(images)  # computed under the previous step (or now, for the first batch of an epoch)
                if nxt is not None:
                    ahead.submit(nxt["image"])  # runs beside this step's decoder forward / backward
            loss = vlm_step_loss(vit_model, vlm_model, adapter, images, input_ids, mask, hf_vit_model, vit_hidden=vit_hidden)
            if grad_sync is not None:
                grad_sync.begin_step()
                # ragged captions: weight this rank's mean by its share of the global batch's target tokens (ddp.py)
                (loss * grad_sync.loss_weight(mask.sum())).backward()
                grad_sync.finish_step()
            else:
                loss.backward()
            total_loss += loss.detach().float()  # stays on the device: no per-step host sync (upstream calls .item())
            clip_grad_norm_(list(vlm_model.parameters()) + list(adapter.parameters()), max_norm=1.0)
            optimizer.step()
            optimizer.zero_grad()
            if val_loader is not None and eval_freq is not None and (step + 1) % eval_freq == 0:
                tr, va = vlm_evaluation(train_loader, val_loader, vit_model, adapter, vlm_model, eval_iter, device, hf_vit_model)
                print(f"Epoch: {epoch}, Step: {step+1}", f"Train loss: {tr:.5f}, Val loss: {va:.5f}")
            if val_loader is None and eval_freq is not None and (step + 1) % eval_freq == 0:
                print(f"Epoch {epoch}, step {step+1}, Loss: {float(total_loss) / (step + 1):.4f}")
        print(f"Epoch {epoch} completed. Average Loss: {float(total_loss) / max(len(train_loader), 1):.4f}")
    return vlm_model, adapter


def _calc_loss_batch_vlm(images, input_ids, text_attention_mask, vit_model, adapter, vlm_model, device, hf_vit_model=True):
    return vlm_step_loss(vit_model, vlm_model, adapter, images.to(device), input_ids.to(device), text_attention_mask.to(device), hf_vit_model)


def calc_loss_loader_vlm(dataloader, vit_model, adapter, vlm_model, device, num_batches=None, hf_vit_model=True):
    if len(dataloader) == 0:
        return float("NaN")
    num_batches = len(dataloader) if num_batches is None else min(num_batches, len(dataloader))
    total = 0.0
    for i, batch in enumerate(dataloader):
        if i >= num_batches:
            break
        total += _calc_loss_batch_vlm(batch["image"], batch["input_ids"], batch["attention_mask"], vit_model, adapter, vlm_model, device, hf_vit_model).item()
    return total / num_batches


def vlm_evaluation(train_loader, val_loader, vit_model, adapter, vlm_model, eval_iter, device, hf_vit_model=True):
    vit_model.eval()
    adapter.eval()
    vlm_model.eval()
    with torch.no_grad():
        tr = calc_loss_loader_vlm(train_loader, vit_model, adapter, vlm_model, device, eval_iter, hf_vit_model)
        va = calc_loss_loader_vlm(val_loader, vit_model, adapter, vlm_model, device, eval_iter, hf_vit_model)
    adapter.train()
    vlm_model.train()
    return tr, va
