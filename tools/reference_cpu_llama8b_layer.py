"""Build container only: the UNMODIFIED reference on CPU on one decoder layer of
Llama-3-8B's dimensions, W4 group_size=128 sym, at the true minibatch 8 x 2048, `device_map="cpu"`, `enable_torch_compile=False`.

Times only `SignRoundQuantizer.quantize_block` (the per-block tuning loop, what bench.py's "step" iterates) and divides by the
iterations run; calibration caching, the fp / quantised-output forwards and model construction are outside the interval.  The
model is a 1-layer random-init `LlamaForCausalLM` (hidden 4096, ffn 14336, 32 heads / 8 kv heads; a small vocabulary, which is not
on the path); nsamples is kept small because the per-iteration cost does not depend on it.

    python tools/reference_cpu_llama8b_layer.py [--iters 10 --nsamples 16] > profiles/r03_reference_cpu_llama8b_layer.json
"""
import argparse
import json
import os
import platform
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
for p in (os.path.join(ROOT, "oracle", "ref_shim"), "/root/reference", os.path.join(ROOT, "tests"), ROOT):
    sys.path.insert(0, p)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--nsamples", type=int, default=16)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--batch-size", type=int, default=8)
    args = ap.parse_args()

    from transformers import LlamaConfig, LlamaForCausalLM

    from auto_round import AutoRound
    from auto_round.algorithms.quantization.sign_round import quantizer as RQ
    from test_pipeline_vs_reference import _Loader, _StubTokenizer

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=14336, num_attention_heads=32, num_key_value_heads=8, num_hidden_layers=1,
                      vocab_size=4096, rope_theta=500000.0, max_position_embeddings=8192, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    model = LlamaForCausalLM(cfg).to(torch.bfloat16).eval()
    tokens = torch.randint(0, 4096, (args.nsamples, args.seqlen), generator=torch.Generator().manual_seed(1))
    os.makedirs("/tmp/cpu_timing_8b", exist_ok=True)
    os.chdir("/tmp/cpu_timing_8b")

    spent = {"qb": 0.0, "calls": 0}
    iter_marks = []
    orig_qb = RQ.SignRoundQuantizer.quantize_block
    orig_loss = RQ.SignRoundQuantizer._get_loss

    def timed_qb(self, *a, **k):
        t0 = time.perf_counter()
        iter_marks.append(t0)
        try:
            return orig_qb(self, *a, **k)
        finally:
            spent["qb"] += time.perf_counter() - t0
            spent["calls"] += 1

    def marked_loss(self, *a, **k):         # one call per minibatch = per iteration here (gradient_accumulate_steps = 1)
        out = orig_loss(self, *a, **k)
        iter_marks.append(time.perf_counter())
        return out

    RQ.SignRoundQuantizer.quantize_block = timed_qb
    RQ.SignRoundQuantizer._get_loss = marked_loss
    t_all = time.perf_counter()
    ar = AutoRound(model, tokenizer=_StubTokenizer(), iters=args.iters, nsamples=args.nsamples, seqlen=args.seqlen,
                   dataset=_Loader(tokens), device_map="cpu", batch_size=args.batch_size, enable_torch_compile=False, scheme="W4A16",
                   seed=42)
    ar.quantize()
    t_all = time.perf_counter() - t_all
    RQ.SignRoundQuantizer.quantize_block = orig_qb
    RQ.SignRoundQuantizer._get_loss = orig_loss

    # forward-to-forward gaps between consecutive loss evaluations = one full iteration (backward + optimizer step + next forward)
    gaps = [b - a for a, b in zip(iter_marks[1:-1], iter_marks[2:])]
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    have = set(line.split(":", 1)[1].split())
                    flags = " ".join(sorted(x for x in have if x.startswith(("amx", "avx512_bf16", "avx512f", "avx2"))))
                    break
    except OSError:
        pass
    s_iter = spent["qb"] / max(args.iters, 1)
    out = dict(what="the REAL reference (auto_round.AutoRound(...).quantize(), device_map='cpu', enable_torch_compile=False) on ONE decoder "
                    "layer of Llama-3-8B's dimensions, W4 group_size=128 sym, minibatch 8x2048; interval = SignRoundQuantizer.quantize_block only",
               model="1-layer LlamaForCausalLM: hidden 4096, ffn 14336, 32 heads / 8 kv heads, random init, bf16", scheme="W4A16 (W4G128 sym)",
               iters=args.iters, nsamples=args.nsamples, seqlen=args.seqlen, batch_size=args.batch_size,
               nproc=os.cpu_count(), threads=torch.get_num_threads(), cpu=platform.processor() or platform.machine(), isa_flags=flags,
               quantize_block_s=spent["qb"], quantize_block_calls=spent["calls"], reference_tuning_s_per_iter=s_iter,
               iteration_gaps_s=[round(g, 2) for g in gaps],
               reference_s_per_block_at_200_iters=200 * s_iter, reference_blocks_per_s=1.0 / (200 * s_iter),
               whole_run_s=t_all, torch=torch.__version__,
               note="kind 'reference': quoted by bench.py as cpu_reference_quoted for the Llama-3-8B headline; the reference tree does not "
                    "exist on the GPU box, so this number is measured in the build container (8 vCPUs) and quoted with its core count")
    print(json.dumps(out))


if __name__ == "__main__":
    main()
