#!/usr/bin/env python
"""Front-door throughput at BASELINE size: a random-init model with Llama-3-8B's layer dimensions (N decoder blocks) goes through
`AutoRound(...).quantize()` and `.save_quantized()` with the reference's default recipe (W4A16 g128, 200 iterations, 128 x 2048
calibration tokens, batch 8).  Shows that calibration capture, block chaining, packing and shard writing around the hot path do
not eat the per-block rate bench.py reports.  GPU box only; one JSON line."""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--layers", type=int, default=3)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--scheme", default="W4A16")
    ap.add_argument("--alg-ext", action="store_true")
    a = ap.parse_args()
    from transformers import LlamaConfig, LlamaForCausalLM

    from auto_round_amd.autoround import AutoRound

    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8,
                      num_hidden_layers=a.layers, vocab_size=128256, rope_theta=500000.0, max_position_embeddings=8192,
                      tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    torch.manual_seed(0)
    with torch.device("cuda"):
        model = LlamaForCausalLM(cfg).to(torch.bfloat16)
    tokens = torch.randint(0, 128256, (128, 2048), generator=torch.Generator().manual_seed(1))
    ar = AutoRound(model, None, scheme=a.scheme, iters=a.iters, nsamples=128, seqlen=2048, batch_size=8, dataset=tokens,
                   enable_alg_ext=a.alg_ext)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ar.quantize()
    torch.cuda.synchronize(); t_q = time.perf_counter() - t0
    out = tempfile.mkdtemp(prefix="ar_e2e_")
    t0 = time.perf_counter()
    ar.save_quantized(out)
    t_s = time.perf_counter() - t0
    size = sum(os.path.getsize(os.path.join(out, f)) for f in os.listdir(out))
    shutil.rmtree(out)
    print(json.dumps({"layers": a.layers, "scheme": a.scheme, "alg_ext": a.alg_ext, "iters": a.iters,
                      "quantize_s": round(t_q, 2), "quantize_s_per_block": round(t_q / a.layers, 3),
                      "save_s": round(t_s, 2), "checkpoint_bytes": size,
                      "block_stats": [{k: r["stats"][k] for k in ("init_loss", "best_loss", "best_iter")} for r in ar.records]}))


if __name__ == "__main__":
    main()
