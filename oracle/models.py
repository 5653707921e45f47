"""Model-level CPU restatements, functional over a state_dict.  TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py).  State-dict key names are the reference's, so the same
dict loads into the reference modules, into ``llm_quest_amd`` modules and into these functions.
"""

import torch
import torch.nn.functional as F

from . import ops


# --------------------------------------------------------------------------- Qwen3 dense decoder
def qwen3_attention(sd, pfx, x, cfg, cos, sin, key_mask=None, position_ids=None):
    """GroupedQueryAttention.forward (qwen/qwen3/qwen3_attention.py:81-150)."""
    b, s, _ = x.shape
    hq, hkv, dh = cfg["n_heads"], cfg["num_kv_groups"], cfg["head_dim"]
    q = F.linear(x, sd[pfx + "w_queries.weight"]).view(b, s, hq, dh).transpose(1, 2)
    k = F.linear(x, sd[pfx + "w_keys.weight"]).view(b, s, hkv, dh).transpose(1, 2)
    v = F.linear(x, sd[pfx + "w_values.weight"]).view(b, s, hkv, dh).transpose(1, 2)
    q = ops.rmsnorm(q, sd[pfx + "q_norm.weight"])  # QK-norm BEFORE RoPE (:108-111)
    k = ops.rmsnorm(k, sd[pfx + "k_norm.weight"])
    q = ops.rope_apply(q, cos, sin, position_ids)
    k = ops.rope_apply(k, cos, sin, position_ids)
    ctx = ops.gqa_attention_core(q, k, v, hq // hkv, key_mask=key_mask, causal=True)
    ctx = ctx.transpose(1, 2).contiguous().view(b, s, hq * dh)
    return F.linear(ctx, sd[pfx + "out_proj.weight"])


def qwen3_block(sd, pfx, x, cfg, cos, sin, key_mask=None, position_ids=None):
    """TransformerBlock.forward (qwen/qwen3/qwen3_transformer_block.py:91-103): pre-norm residual block."""
    h = ops.rmsnorm(x, sd[pfx + "norm1.weight"])
    x = qwen3_attention(sd, pfx + "att.", h, cfg, cos, sin, key_mask, position_ids) + x
    h = ops.rmsnorm(x, sd[pfx + "norm2.weight"])
    x = ops.swiglu_ffn(h, sd[pfx + "ffn.lin1.weight"], sd[pfx + "ffn.lin_gate.weight"], sd[pfx + "ffn.lin2.weight"]) + x
    return x


def qwen3_forward(sd, cfg, tokens=None, key_mask=None, position_ids=None, inputs_embeds=None):
    """Qwen3Model.forward (qwen/qwen3/qwen3_model.py:60-94).  ``inputs_embeds`` is the embedded-input entry the
    composed VLM needs (SURVEY.md section 8c harness: blocks -> final_norm -> tied out_head)."""
    cos, sin = ops.rope_tables(cfg["rope_base"], cfg["head_dim"], cfg["context_length"])
    x = F.embedding(tokens, sd["emb_dict.weight"]) if inputs_embeds is None else inputs_embeds
    for i in range(cfg["n_layers"]):
        x = qwen3_block(sd, f"trf_blocks.{i}.", x, cfg, cos, sin, key_mask, position_ids)
    x = ops.rmsnorm(x, sd["final_norm.weight"])
    head = sd["out_head.weight"] if "out_head.weight" in sd else sd["emb_dict.weight"]
    return F.linear(x, head)


# --------------------------------------------------------------------------- ViT encoder
def vit_attention(sd, pfx, x, n_heads, att_mul=None):
    """ViTMultiHeadAttention.forward (multimodal/vision_transformer/vit_attention.py:43-91).  ``att_mul`` stands for the
    reference's ``self.dropout(att_weights)`` (:79) with the mask given: a (b, heads, s, s) tensor of 0 or 1 / (1 - p)."""
    b, s, d = x.shape
    dh = d // n_heads

    def proj(name):
        return F.linear(x, sd[pfx + name + ".weight"], sd.get(pfx + name + ".bias")).view(b, s, n_heads, dh).transpose(1, 2)

    if att_mul is None:
        ctx = ops.full_attention_core(proj("w_queries"), proj("w_keys"), proj("w_values"))
    else:
        q, k, v = proj("w_queries"), proj("w_keys"), proj("w_values")
        ctx = (torch.softmax((q @ k.mT) * (dh**-0.5), dim=-1) * att_mul) @ v
    ctx = ctx.transpose(1, 2).contiguous().view(b, s, d)
    return F.linear(ctx, sd[pfx + "out_proj.weight"], sd[pfx + "out_proj.bias"])


def vit_block(sd, pfx, x, n_heads, drop=None):
    """ViTTransformerBlock.forward (vit_transformer_block.py:102-127).  ``drop`` = None (drop_rate 0 / eval) or a dict of given
    dropout multipliers: "att" (b, heads, s, s), "proj" and "ffn" (b, s, d) for the two ``self.dropout`` sites (:117, :124)."""
    drop = drop or {}
    h = ops.layernorm_sigma_eps(x, sd[pfx + "ln_1.scale"], sd[pfx + "ln_1.shift"])
    a = vit_attention(sd, pfx + "att.", h, n_heads, drop.get("att"))
    x = (a * drop["proj"] if "proj" in drop else a) + x
    h = ops.layernorm_sigma_eps(x, sd[pfx + "ln_2.scale"], sd[pfx + "ln_2.shift"])
    h = F.linear(h, sd[pfx + "ffn.layers.0.weight"], sd[pfx + "ffn.layers.0.bias"])
    h = F.linear(ops.gelu_erf(h), sd[pfx + "ffn.layers.2.weight"], sd[pfx + "ffn.layers.2.bias"])
    return (h * drop["ffn"] if "ffn" in drop else h) + x


def vit_forward(sd, cfg, img, output_hidden_states=False, drop=None):
    """ViTModel.forward (multimodal/vision_transformer/vit_model.py:134-160).  ``drop`` = None for eval / drop_rate 0, or
    {"embed": (b, s, d), "blocks": [per-block dict as in vit_block]} = the train-mode dropout sites with their masks given."""
    x = ops.patch_embed(
        img,
        sd["patch_embedding.conv_proj.weight"],
        sd["patch_embedding.conv_proj.bias"],
        sd["patch_embedding.cls_token"],
        cfg["patch_size"],
    )
    x = x + sd["pos_embedding"]
    if drop is not None:
        x = x * drop["embed"]
    for i in range(cfg["n_layers"]):
        x = vit_block(sd, f"transformer_blocks.{i}.", x, cfg["n_heads"], None if drop is None else drop["blocks"][i])
    x = ops.layernorm_sigma_eps(x, sd["final_ln.scale"], sd["final_ln.shift"])
    if output_hidden_states:
        return x
    return F.linear(x[:, 0], sd["classifier.weight"], sd["classifier.bias"])


# --------------------------------------------------------------------------- adapter + fusion
def adapter_forward(sd, x):
    """ViTAdapter.forward (multimodal/vision_transformer/vit_engine.py:32-59); "simple" or "ffn" by keys."""
    if "adapter.weight" in sd:
        return F.linear(x, sd["adapter.weight"], sd.get("adapter.bias"))
    h = F.linear(x, sd["adapter.0.weight"], sd.get("adapter.0.bias"))
    h = F.gelu(h)  # nn.GELU() = erf form (:50)
    return F.linear(h, sd["adapter.3.weight"], sd.get("adapter.3.bias"))


def vlm_forward_loss(vit_sd, vit_cfg, ad_sd, llm_sd, llm_cfg, image, input_ids, text_mask):
    """Composed early-fusion train-step forward for BASELINE config 4 (SURVEY.md section 8c harness, built from
    multimodal/vlm_engine.py:94-128 with Qwen3 in place of GPT-2): frozen ViT hidden states -> adapter (bf16) ->
    cat([vision, tok_emb]) -> Qwen3 blocks with causal | ~key_mask -> final_norm -> tied head -> vlm_loss.
    Returns (loss, logits, fused_embeddings)."""
    with torch.no_grad():
        hid = vit_forward(vit_sd, vit_cfg, image, output_hidden_states=True)
    ad_dtype = next(iter(ad_sd.values())).dtype
    vis = adapter_forward(ad_sd, hid.to(ad_dtype))
    txt = F.embedding(input_ids, llm_sd["emb_dict.weight"])
    fused = torch.cat([vis.to(txt.dtype), txt], dim=1)
    n_v = vis.shape[1]
    key_mask = torch.cat([torch.ones(image.shape[0], n_v, dtype=torch.bool), text_mask.to(torch.bool)], dim=1)
    logits = qwen3_forward(llm_sd, llm_cfg, key_mask=key_mask, inputs_embeds=fused)
    return ops.vlm_loss(logits, input_ids, text_mask, n_v), logits, fused


# --------------------------------------------------------------------------- GPT-2 (config-1 plumbing)
def gpt2_forward(sd, cfg, tokens, key_mask=None):
    """GPTModel.forward, drop_rate=0, no cache (gpt/gpt_model.py:43-118, gpt_attention.py:155-215,
    gpt_transformer_block.py:125-146)."""
    b, s = tokens.shape
    x = F.embedding(tokens, sd["emb_dict.weight"]) + sd["pos_emb_dict.weight"][:s]
    nh = cfg["n_heads"]
    dh = cfg["emb_dim"] // nh
    for i in range(cfg["n_layers"]):
        p = f"trf_blocks.{i}."
        h = ops.layernorm_sigma_eps(x, sd[p + "ln_1.scale"], sd[p + "ln_1.shift"])

        def proj(name):
            return F.linear(h, sd[p + "att." + name + ".weight"], sd.get(p + "att." + name + ".bias")).view(b, s, nh, dh).transpose(1, 2)

        ctx = ops.gqa_attention_core(proj("w_queries"), proj("w_keys"), proj("w_values"), 1, key_mask=key_mask, causal=True)
        ctx = ctx.transpose(1, 2).contiguous().view(b, s, nh * dh)
        x = F.linear(ctx, sd[p + "att.out_proj.weight"], sd[p + "att.out_proj.bias"]) + x
        h = ops.layernorm_sigma_eps(x, sd[p + "ln_2.scale"], sd[p + "ln_2.shift"])
        h = F.linear(h, sd[p + "ffn.layers.0.weight"], sd[p + "ffn.layers.0.bias"])
        x = F.linear(ops.gelu_erf(h), sd[p + "ffn.layers.2.weight"], sd[p + "ffn.layers.2.bias"]) + x
    x = ops.layernorm_sigma_eps(x, sd["final_ln.scale"], sd["final_ln.shift"])
    return F.linear(x, sd["out.weight"])
