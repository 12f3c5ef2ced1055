#!/usr/bin/env python
"""(test infrastructure; lives under tests/ because it uses the oracle)
Whole-model comparison of the two ENGINES inside the same driver: the model saved by tools/e2e_quality.py --save-model is
quantised block by block with (a) the HIP path (SignRoundQuantizer) and (b) oracle/torch_ref.tune_block -- the pinned torch
restatement of the reference loop -- using the same captured block inputs, token mask, seeds and chaining, and the held-out
perplexity of both is reported.  Separates "does the hot path differ" from "does the front door differ" when comparing with
a reference run (tools/e2e_reference_cpu.py).  usage: python tests/e2e_engine_compare.py MODEL.pt [--amp 0|1]"""
import argparse
import copy
import json
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import transformers

from auto_round_amd.autoround import AutoRound
from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
from oracle import torch_ref as tr


@torch.no_grad()
def perplexity(model, tokens, bs=32):
    nll, cnt = 0.0, 0
    for b0 in range(0, tokens.shape[0], bs):
        t = tokens[b0:b0 + bs].cuda()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(input_ids=t, use_cache=False).logits
        nll += float(torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, logits.shape[-1]), t[:, 1:].reshape(-1),
                                                       reduction="sum"))
        cnt += t[:, 1:].numel()
    return math.exp(nll / cnt)


def _reference_mask(S, device):
    """causal AND (key != last token), boolean in transformers >= 5, cast to bf16 by the reference's input cache"""
    m = torch.tril(torch.ones(S, S, device=device))
    m[:, -1] = 0
    return m.to(torch.bfloat16).reshape(1, 1, S, S)


def fwd(blk, x, others):
    out = blk(x, **others)
    return out[0] if isinstance(out, (tuple, list)) else out


def run_torch_ref(ar, blocks, x0, others, ids, iters, bs, amp):
    """the facade's block loop (model_tuner.tune_blocks) with torch_ref as the engine"""
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, sdpa_backend="auto", amp=amp), device="cuda")
    fp_in, q_in = x0, None
    for blk in blocks:
        fp_out = q.forward_all(blk, fp_in, others)
        xin = q_in if q_in is not None else fp_in
        tr.tune_block(blk, xin, fp_out, others, iters=iters, batch_size=bs, forward=fwd, input_ids=ids, amp=amp)
        q_in = q.forward_all(blk, xin, others)
        fp_in = fp_out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("model")
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--seed", type=int, default=42)
    ap.add_argument("--schemes", default="W4A16 g32;W2A16 g32")
    ap.add_argument("--skip-torch-ref", action="store_true")
    ap.add_argument("--reference-mask", action="store_true",
                    help="feed the blocks the attention mask the reference's front door ends up caching with transformers >= 5: the "
                         "boolean causal mask cast to the amp dtype (calibration/inputs.py:100-107), i.e. a 0/1 ADDITIVE bias -- "
                         "reproduces the reference's CPU numbers (tools/e2e_reference_cpu.py)")
    a = ap.parse_args()
    from transformers import LlamaConfig, LlamaForCausalLM

    blob = torch.load(a.model)
    cfg = LlamaConfig(**{k: v for k, v in blob["config"].items() if k not in ("architectures", "model_type", "transformers_version", "dtype")})
    cfg._attn_implementation = "sdpa"
    base = LlamaForCausalLM(cfg).to(torch.bfloat16)
    base.load_state_dict(blob["state_dict"])
    base = base.cuda().eval()
    res = {"ppl_bf16": round(perplexity(base, blob["held"]), 4), "iters": a.iters, "reference_mask": bool(a.reference_mask), "schemes": {}}
    all_schemes = {"W4A16 g32": dict(scheme="W4A16", group_size=32), "W2A16 g32": dict(scheme="W2A16G32"),
                   "W2A16 g32 alg_ext": dict(scheme="W2A16G32", enable_alg_ext=True), "MXFP4 alg_ext": dict(scheme="MXFP4", enable_alg_ext=True),
                   "W4A8 g32": dict(bits=4, act_bits=8, group_size=32, act_group_size=32, sym=True),
                   "MXFP4": dict(scheme="MXFP4"), "NVFP4": dict(scheme="NVFP4"), "INT8 W8A8": dict(scheme="INT8")}
    for name in [x.strip() for x in a.schemes.split(";")]:
        kw = all_schemes[name]
        row = {}
        m = copy.deepcopy(base)
        ar0 = AutoRound(m, None, nsamples=128, seqlen=128, batch_size=8, dataset=blob["calib"], iters=a.iters, seed=a.seed, **kw)
        if a.reference_mask:
            cap0 = ar0._capture_block0_inputs

            def cap_with_ref_mask(blocks, tokens):
                x0, others = cap0(blocks, tokens)
                S = x0.shape[1]
                others["attention_mask"] = _reference_mask(S, x0.device)
                return x0, others

            ar0._capture_block0_inputs = cap_with_ref_mask
            ar0.config.sdpa_backend = "auto"
        ar0.quantize()
        row["hip_engine"] = round(perplexity(m, blob["held"]), 4)
        row["hip_block_losses"] = [[r["stats"]["init_loss"], r["stats"]["best_loss"]] for r in ar0.records]
        if a.skip_torch_ref:
            res["schemes"][name] = row
            print(name, row, file=sys.stderr)
            continue
        # same driver, torch_ref engine
        m = copy.deepcopy(base)
        ar = AutoRound(m, None, nsamples=128, seqlen=128, batch_size=8, dataset=blob["calib"], iters=a.iters, seed=a.seed, **kw)
        ar.config.sdpa_backend = "auto"
        transformers.set_seed(a.seed)
        from auto_round_amd.autoround import get_block_names
        from auto_round_amd.schemes import apply_scheme

        m = m.cuda().eval()
        for p in m.parameters():
            p.requires_grad_(False)
        names = max(get_block_names(m), key=len)
        blocks = [m.get_submodule(n) for n in names]
        for b in blocks:
            apply_scheme(b, ar.scheme)
        tokens = ar._calibration_tokens()
        ids = tokens.clone(); ids[:, -1] = -100
        x0, others = ar._capture_block0_inputs(blocks, tokens)
        if a.reference_mask:
            S = x0.shape[1]
            others["attention_mask"] = _reference_mask(S, x0.device)
        run_torch_ref(ar, blocks, x0, others, ids, a.iters, 8, True)
        row["torch_ref_engine"] = round(perplexity(m, blob["held"]), 4)
        res["schemes"][name] = row
        print(name, row, file=sys.stderr)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
