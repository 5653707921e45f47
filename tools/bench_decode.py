"""Decode throughput of Qwen3-0.6B through the KV cache: prefill P tokens, then N one-token steps.
Bound: HBM -- every step streams all weights once (1.19 GB bf16 incl. the tied 151 936 x 1024 head) plus the cache.
GPU box only:  python tools/bench_decode.py [--batch 1] [--prompt 512] [--steps 64] [--graph]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from llm_quest_amd.config import qwen3_config_creator
from llm_quest_amd.qwen.qwen3.qwen3_model import Qwen3Model
from llm_quest_amd.utils import KVCache
from llm_quest_amd import ops_decode

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=1); ap.add_argument("--prompt", type=int, default=512); ap.add_argument("--steps", type=int, default=64)
ap.add_argument("--graph", action="store_true", help="replay one captured hipGraph per token (ops_decode.GraphDecoder)")
a = ap.parse_args()
dev = "cuda"
torch.manual_seed(123)
with torch.device(dev):
    m = Qwen3Model(dict(qwen3_config_creator("0.6B"), context_length=2048)).eval()
ids = torch.randint(0, 151_936, (a.batch, a.prompt), device=dev)
with torch.inference_mode():
    kv = KVCache(num_layers=28, prompt_len=a.prompt, context_len=2048)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    logits = m(ids, kv_cache=kv)[:, -1]
    torch.cuda.synchronize(); t_prefill = time.perf_counter() - t0
    pos = torch.tensor([[a.prompt]], device=dev)
    tok = ops_decode.argmax_rows(logits).unsqueeze(-1)
    for _ in range(4):  # warm-up steps
        tok = ops_decode.argmax_rows(m(tok, kv_cache=kv, position_ids=pos).squeeze(1)).unsqueeze(-1); pos += 1
    if a.graph:
        dec = ops_decode.GraphDecoder(m, kv, tok, a.steps + 4)
        for i in range(3):
            torch.cuda.synchronize(); tt = time.perf_counter()
            dec.step()
            torch.cuda.synchronize(); print(f"  graph step {i}: {(time.perf_counter() - tt)*1e3:.1f} ms")
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            dec.graph.replay(); dec._advance_host()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    else:
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(a.steps):
            tok = ops_decode.argmax_rows(m(tok, kv_cache=kv, position_ids=pos).squeeze(1)).unsqueeze(-1); pos += 1
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
wbytes = sum(p.numel() * p.element_size() for p in m.parameters())
ctx = a.prompt + 4 + a.steps // 2
cbytes = 28 * 2 * a.batch * ctx * 8 * 128 * 2
print(f"prefill {a.prompt} tok x {a.batch}: {t_prefill*1e3:.1f} ms (first call, includes arena build)")
mode = " (hipGraph)" if a.graph else ""
print(f"decode{mode} B={a.batch} ctx~{ctx}: {dt*1e3:.3f} ms/step  {a.batch/dt:.0f} tok/s   algorithmic bytes/step {((wbytes+cbytes)/1e9):.3f} GB -> {(wbytes+cbytes)/dt/1e12:.2f} TB/s = {(wbytes+cbytes)/dt/8e12*100:.1f} % of 8 TB/s HBM")
if a.graph:
    dec.close()
