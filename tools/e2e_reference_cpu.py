#!/usr/bin/env python
"""Build container only: run the REFERENCE (intel/auto-round, CPU path) on the model that tools/e2e_quality.py trained on the
MI355X, with the same calibration tokens and settings, and report the held-out perplexity next to ours.
usage: python tools/e2e_reference_cpu.py gpurun_out/e2e_model.pt gpurun_out/e2e_seeds.json   (the latter from e2e_quality.py --load-model) > profiles/...json"""
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(ROOT, "oracle", "ref_shim"))
sys.path.insert(0, "/root/reference")
import torch  # noqa: E402


@torch.no_grad()
def perplexity(model, tokens, bs=32):
    nll, cnt = 0.0, 0
    for b0 in range(0, tokens.shape[0], bs):
        t = tokens[b0:b0 + bs]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            logits = model(input_ids=t, use_cache=False).logits
        nll += float(torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, logits.shape[-1]), t[:, 1:].reshape(-1),
                                                       reduction="sum"))
        cnt += t[:, 1:].numel()
    return math.exp(nll / cnt)


class StubTokenizer:
    pad_token_id = None
    pad_token = None

    def save_pretrained(self, *a, **k):
        pass


class Loader:
    batch_size = 1

    def __init__(self, tokens):
        self.tokens = tokens

    def __iter__(self):
        for r in self.tokens:
            yield r.reshape(1, -1)


def main():
    from transformers import LlamaConfig, LlamaForCausalLM

    from auto_round import AutoRound

    blob = torch.load(sys.argv[1])
    ours = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
    cfg = LlamaConfig(**{k: v for k, v in blob["config"].items() if k not in ("architectures", "model_type", "transformers_version", "dtype")})
    seeds = [int(x) for x in next(iter(ours["seeds"].values())).keys()]
    out = {"ppl_bf16_cpu": None, "ours_ppl_bf16_gpu": ours["ppl_bf16"], "iters": ours["iters"], "schemes": {}}
    os.chdir("/tmp")
    all_schemes = {"W4A16 g32": dict(scheme="W4A16", group_size=32), "W2A16 g32": dict(scheme="W2A16G32"),
                   "W2A16 g32 asym": dict(scheme="W2A16G32", sym=False), "W2A16 g32 alg_ext": dict(scheme="W2A16G32", enable_alg_ext=True), "MXFP4 alg_ext": dict(scheme="MXFP4", enable_alg_ext=True),
                   "W4A8 g32": dict(bits=4, act_bits=8, group_size=32, act_group_size=32, sym=True),
                   "MXFP4": dict(scheme="MXFP4"), "NVFP4": dict(scheme="NVFP4"),
                   "INT8 W8A8": dict(scheme="INT8")}
    for name in ours["seeds"]:
        kw = all_schemes[name]
        ref = {}
        for seed in seeds:
            model = LlamaForCausalLM(cfg).to(torch.bfloat16)
            model.load_state_dict(blob["state_dict"])
            model.eval()
            if out["ppl_bf16_cpu"] is None:
                out["ppl_bf16_cpu"] = round(perplexity(model, blob["held"]), 4)
            ar = AutoRound(model, tokenizer=StubTokenizer(), iters=ours["iters"], nsamples=128, seqlen=blob["calib"].shape[1],
                           dataset=Loader(blob["calib"]), device_map="cpu", batch_size=8, enable_torch_compile=False, seed=seed, **kw)
            qmodel, _ = ar.quantize()
            ref[str(seed)] = round(perplexity(qmodel, blob["held"]), 4)
        out["schemes"][name] = {"reference_cpu_signround": ref, "ours_gpu_signround": ours["seeds"][name]}
        print(name, out["schemes"][name], file=sys.stderr)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
