"""Public call signatures of a Python source tree, read with ``ast`` (nothing is imported or executed).

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  ``oracle/gen_golden.py`` runs ``collect`` over the reference's modules on
the hot path and commits the result as ``tests/golden/signatures.json`` -- names and argument lists,
data only; ``tests/test_signatures_cpu.py`` runs the same function over ``llm_quest_amd`` and compares.
"""

import ast

# reference modules (relative to the package root) that make up the drop-in boundary of SURVEY.md section 8(b)
MODULES = [
    "common/rope.py", "common/buffers.py", "engine.py", "generate.py", "utils.py", "dataset.py",
    "multimodal/vlm_engine.py", "multimodal/vlm_generation.py",
    "multimodal/vision_transformer/vit_attention.py", "multimodal/vision_transformer/vit_engine.py",
    "multimodal/vision_transformer/vit_model.py", "multimodal/vision_transformer/vit_transformer_block.py",
    "qwen/qwen3/qwen3_attention.py", "qwen/qwen3/qwen3_model.py", "qwen/qwen3/qwen3_transformer_block.py",
    "qwen/qwen3/qwen3_weight_loading.py",
    "qwen/qwen3_5/qwen3_5_text_model.py", "qwen/qwen3_5/qwen3_5_vision_model.py", "qwen/qwen3_5/qwen3_5_vlm_model.py",
    "qwen/qwen3_5/qwen3_5_weight_loading.py", "qwen/qwen3_next/qwen3_next_attention.py",
    "gpt/gpt_model.py",
]

# symbols of those modules that SURVEY.md section 2 / 8 mark OUT OF SCOPE (not on the hot path): left out of the fixture, with the reason
OUT_OF_SCOPE = {
    "common/buffers.py": {"GlobalBuffers.get_swa_buffers": "sliding-window attention (Gemma family)", "GlobalBuffers.get_swa_mask": "sliding-window attention"},
    "qwen/qwen3/qwen3_model.py": {"Qwen3MoEModel": "MoE variant"},
    "qwen/qwen3/qwen3_transformer_block.py": {"MoETransformerBlock": "MoE variant"},
    "qwen/qwen3_next/qwen3_next_attention.py": {"GatedDeltaNet": "unfused teaching variant; config 5 uses FusedGatedDeltaNet", "l2_norm_official": "HF cross-check helper"},
    "generate.py": {"generate_simple_loop": "no-sampling teaching loop", "generate_batched_loop": "batched loop without a KV cache (row f4 is KV-cache inference)"},
    "utils.py": {
        "time_it": "decorator", "text_to_ids": "tokenizer glue", "ids_to_text": "tokenizer glue", "alpaca_prompt_format": "SFT prompt text",
        "alpaca_deepseek_format": "SFT prompt text", "ResponseExtractor": "alignment / RL", "EntropyFilteredTokens": "alignment / RL",
        "CheckpointEvaluator": "alignment / RL", "SinkhornKnopp": "mHC experiments", "BirkhoffvonNeumann": "mHC experiments",
        "test_generation_with_weights": "needs the hub tokenizer",
    },
    "dataset.py": {"*": "only MultimodalDataset is on the path (row f3)", "keep": ["MultimodalDataset"]},
    "gpt/gpt_model.py": {},
}

_KEEP_PRIVATE = {"__init__", "__call__", "_calc_loss_batch"}


def _sig(fn):
    a = fn.args

    def one(arg, default):
        return arg.arg if default is None else f"{arg.arg}={ast.unparse(default)}"

    pos = a.posonlyargs + a.args
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    parts = [one(p, d) for p, d in zip(pos, defaults)]
    if a.vararg:
        parts.append("*" + a.vararg.arg)
    elif a.kwonlyargs:
        parts.append("*")
    parts += [one(p, d) for p, d in zip(a.kwonlyargs, a.kw_defaults)]
    if a.kwarg:
        parts.append("**" + a.kwarg.arg)
    return "(" + ", ".join(parts) + ")"


def collect(path, skip=None):
    """{symbol: signature} for the top-level functions and the methods of top-level classes of one source file; annotations are
    dropped, defaults are kept as source text."""
    skip = skip or {}
    keep_only = skip.get("keep") if "*" in skip else None
    tree = ast.parse(open(path).read())
    out = {}
    for node in tree.body:
        if isinstance(node, ast.FunctionDef):
            if node.name.startswith("_") or node.name in skip or (keep_only is not None and node.name not in keep_only):
                continue
            out[node.name] = _sig(node)
        elif isinstance(node, ast.ClassDef):
            if node.name.startswith("_") or node.name in skip or (keep_only is not None and node.name not in keep_only):
                continue
            for sub in node.body:
                if isinstance(sub, ast.FunctionDef) and (not sub.name.startswith("_") or sub.name in _KEEP_PRIVATE):
                    key = f"{node.name}.{sub.name}"
                    if key not in skip:
                        out[key] = _sig(sub)
    return out


def collect_tree(root):
    import os

    return {m: collect(os.path.join(root, m), OUT_OF_SCOPE.get(m)) for m in MODULES if os.path.exists(os.path.join(root, m))}
