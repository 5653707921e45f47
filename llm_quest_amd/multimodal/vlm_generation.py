"""Caption generation with the early-fusion VLM -- API of ``llm_quest/multimodal/vlm_generation.py`` (``vlm_generate_loop``).

Upstream re-runs the whole growing sequence (197 vision rows + generated tokens) through the language model for every token.  Here
the three native pieces of the path are composed instead: the image goes through the GPU input pipeline (Pillow-exact resize +
ToTensor + Normalize, ``llm_quest_amd/dataset.py``), the fused sequence is PREFILLED once into a ``KVCache``
(``Qwen3Model.forward(..., input_embedded=True, kv_cache=...)``), and every further token is a one-token decode step -- one captured
hipGraph per token when decoding greedily (``temp == 0``).  Same arguments and return value as the reference; ``vlm_model`` is the
native ``Qwen3Model`` (the reference wires GPT-2 here; Qwen3 is the language model of BASELINE config 4).
"""

import torch

from llm_quest_amd import ops_decode
from llm_quest_amd.dataset import IMAGENET_MEAN, IMAGENET_STD, _as_rgb_u8, _DeviceTables, image_transform_into
from llm_quest_amd.generate import sampling
from llm_quest_amd.utils import KVCache


def preprocess_image(image, image_size=224, device="cuda"):
    """PIL image / uint8 (H, W, 3) array -> fp32 (1, 3, s, s) on the device: resize, ToTensor, ImageNet Normalize (reference :46-53)."""
    dev = torch.device(device)
    raw = torch.from_numpy(_as_rgb_u8(image)).to(dev)
    out = torch.empty((3, image_size, image_size), dtype=torch.float32, device=dev)
    mean = torch.tensor(IMAGENET_MEAN, dtype=torch.float32, device=dev)
    std = torch.tensor(IMAGENET_STD, dtype=torch.float32, device=dev)
    image_transform_into(raw, out, _DeviceTables(dev), image_size, mean, std)
    return out.unsqueeze(0)


def vlm_generate_ids(image, vit_model, adapter, vlm_model, eos_token_id, max_gen=70, context_length=300, top_k=None, top_p=0.9, temp=0.8,
                     device="cuda", hf_vit_model=True, image_size=224):
    """The token ids ``vlm_generate_loop`` decodes into the caption (a list of ints, the eos included if it was produced)."""
    vit_model.eval().to(device)
    adapter.eval().to(device)
    vlm_model.eval().to(device)
    with torch.inference_mode():
        image_tensor = preprocess_image(image, image_size, device)
        hidden = vit_model(image_tensor, output_hidden_states=True) if not hf_vit_model else vit_model(image_tensor).last_hidden_state
        vision = adapter(hidden.to(vlm_model.emb_dict.weight.dtype))
        vision = vision[:, -context_length:]
        n_vis = vision.shape[1]
        if n_vis + max_gen > context_length:
            max_gen = max(context_length - n_vis, 0)  # the cache does not slide; upstream would start truncating the vision rows here
        kv = KVCache(num_layers=len(vlm_model.trf_blocks), prompt_len=n_vis, context_len=context_length)
        logits = vlm_model(vision.contiguous(), input_embedded=True, kv_cache=kv)[:, -1, :]
        ids, dec = [], None
        pos = torch.tensor([[n_vis]], dtype=torch.long, device=device)
        nxt = sampling(logits, top_k, top_p, None, temp) if max_gen > 0 else None
        try:
            while nxt is not None:
                tok = int(nxt.item())
                ids.append(tok)
                if tok == eos_token_id or len(ids) == max_gen:
                    break
                if temp == 0.0:  # greedy: the captured decode step samples on the device and hands back the next token
                    if dec is None:
                        dec = ops_decode.GraphDecoder(vlm_model, kv, nxt, max_gen)
                    nxt = dec.step()
                else:
                    logits = vlm_model(nxt, kv_cache=kv, position_ids=pos).squeeze(1)
                    pos += 1
                    nxt = sampling(logits, top_k, top_p, None, temp)
        finally:
            if dec is not None:
                dec.close()
    return ids


def vlm_generate_loop(image, vit_model, adapter, vlm_model, tokenizer, max_gen=70, context_length=300, top_k=None, top_p=0.9, temp=0.8,
                      device="cuda", hf_vit_model=True, image_size=224):
    """Caption string for ``image`` (reference: vlm_generation.py:8-96)."""
    ids = vlm_generate_ids(image, vit_model, adapter, vlm_model, tokenizer.eos_token_id, max_gen, context_length, top_k, top_p, temp, device,
                           hf_vit_model, image_size)
    return tokenizer.decode(ids, skip_special_tokens=True)
