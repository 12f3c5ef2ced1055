#!/usr/bin/env python
"""End-to-end sanity of the whole path on a model that has actually learned something (there are no checkpoints offline):
train a small Llama on a synthetic order-2 Markov language for a minute, then compare held-out perplexity of
  bf16  |  RTN (iters=0)  |  SignRound (iters=200)  |  SignRound + algorithm extension
for several schemes, all through `auto_round_amd.autoround.AutoRound`.  GPU box only; writes one JSON line."""
import argparse
import copy
import json
import math
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch


def markov_tokens(n_seq, seqlen, vocab, seed):
    """Order-2 Markov chain with a sparse, peaked transition table (entropy ~1.5 nats/token)."""
    g = torch.Generator().manual_seed(1234)                 # the LANGUAGE is fixed; `seed` only picks the samples
    k = 4
    nxt = torch.randint(0, vocab, (vocab, vocab, k), generator=g)
    p = torch.softmax(torch.randn(vocab, vocab, k, generator=g) * 1.5, dim=-1)
    g2 = torch.Generator().manual_seed(seed)
    out = torch.empty(n_seq, seqlen, dtype=torch.long)
    out[:, :2] = torch.randint(0, vocab, (n_seq, 2), generator=g2)
    for t in range(2, seqlen):
        a, b = out[:, t - 2], out[:, t - 1]
        choice = torch.multinomial(p[a, b], 1, generator=g2).squeeze(1)
        out[:, t] = nxt[a, b, choice]
    return out


@torch.no_grad()
def perplexity(model, tokens, bs=32):
    model.eval()
    nll, cnt = 0.0, 0
    for b0 in range(0, tokens.shape[0], bs):
        t = tokens[b0:b0 + bs].cuda()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(input_ids=t, use_cache=False).logits
        loss = torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, logits.shape[-1]), t[:, 1:].reshape(-1), reduction="sum")
        nll += float(loss); cnt += t[:, 1:].numel()
    return math.exp(nll / cnt)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--train-steps", type=int, default=1500)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--load-model", default=None, help="skip training: use the model / tokens saved by --save-model and only run "
                                                       "SignRound for W4A16 g32 / W2A16 g32 / W2A16 g32 asym over --seeds")
    ap.add_argument("--seeds", default="42,43,44")
    ap.add_argument("--save-model", default=None, help="write the trained bf16 state dict + calibration / held-out tokens here")
    a = ap.parse_args()
    from transformers import LlamaConfig, LlamaForCausalLM

    from auto_round_amd.autoround import AutoRound

    vocab, seqlen = 64, 128
    if a.load_model:
        blob = torch.load(a.load_model)
        cfg = LlamaConfig(**{k: v for k, v in blob["config"].items() if k not in ("architectures", "model_type", "transformers_version", "dtype")})
        cfg._attn_implementation = "sdpa"
        model = LlamaForCausalLM(cfg).to(torch.bfloat16)
        model.load_state_dict(blob["state_dict"])
        model = model.cuda().eval()
        res = {"ppl_bf16": round(perplexity(model, blob["held"]), 4), "iters": a.iters, "seeds": {}}
        for name, kw in (("W4A16 g32", dict(scheme="W4A16", group_size=32)), ("W2A16 g32", dict(scheme="W2A16G32")),
                         ("W2A16 g32 asym", dict(scheme="W2A16G32", sym=False))):
            res["seeds"][name] = {}
            for seed in [int(x) for x in a.seeds.split(",")]:
                m = copy.deepcopy(model)
                AutoRound(m, None, nsamples=128, seqlen=seqlen, batch_size=8, dataset=blob["calib"], iters=a.iters, seed=seed, **kw).quantize()
                res["seeds"][name][str(seed)] = round(perplexity(m, blob["held"]), 4)
            print(name, res["seeds"][name], file=sys.stderr)
        print(json.dumps(res))
        return
    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=256, intermediate_size=768, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=4,
                      vocab_size=vocab, max_position_embeddings=seqlen, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    model = LlamaForCausalLM(cfg).cuda()
    train = markov_tokens(32768, seqlen, vocab, seed=1)      # 4 M tokens for 4096 contexts: learnable, not memorisable
    held = markov_tokens(512, seqlen, vocab, seed=2)
    calib = markov_tokens(128, seqlen, vocab, seed=3)
    opt = torch.optim.AdamW(model.parameters(), lr=3e-3, weight_decay=0.01)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=3e-3, total_steps=a.train_steps)
    t0 = time.time()
    model.train()
    for step in range(a.train_steps):
        idx = torch.randint(0, train.shape[0], (64,))
        t = train[idx].cuda()
        with torch.autocast("cuda", dtype=torch.bfloat16):
            logits = model(input_ids=t, use_cache=False).logits
        loss = torch.nn.functional.cross_entropy(logits[:, :-1].float().reshape(-1, vocab), t[:, 1:].reshape(-1))
        opt.zero_grad(set_to_none=True); loss.backward(); opt.step(); sched.step()
    model = model.to(torch.bfloat16).eval()
    if a.save_model:
        torch.save({"state_dict": {k: v.cpu() for k, v in model.state_dict().items()}, "calib": calib, "held": held,
                    "config": cfg.to_dict()}, a.save_model)
    res = {"train_s": round(time.time() - t0, 1), "train_loss_last": round(float(loss.detach()), 4), "ppl_bf16": round(perplexity(model, held), 4),
           "model": "Llama 4x256 (ffn 768), vocab 64, synthetic order-2 Markov language (4096 contexts x 4 continuations)", "iters": a.iters, "schemes": {}}
    for name, kw in (("W4A16 g32", dict(scheme="W4A16", group_size=32)), ("W3A16 g32", dict(scheme="W3A16", group_size=32)),
                     ("W2A16 g32", dict(scheme="W2A16G32")), ("W2A16 g32 asym", dict(scheme="W2A16G32", sym=False)),
                     ("MXFP4", dict(scheme="MXFP4")), ("NVFP4", dict(scheme="NVFP4")), ("INT8 W8A8", dict(scheme="INT8"))):
        row = {}
        for mode, extra in (("rtn", dict(iters=0)), ("signround", dict(iters=a.iters)),
                            ("signround_alg_ext", dict(iters=a.iters, enable_alg_ext=True))):
            if mode == "signround_alg_ext" and kw.get("sym") is False:
                continue                                   # the extension is sym-only (the reference falls back as well)
            m = copy.deepcopy(model)
            t1 = time.time()
            AutoRound(m, None, nsamples=128, seqlen=seqlen, batch_size=8, dataset=calib, **kw, **extra).quantize()
            row[mode] = round(perplexity(m, held), 4)
            row[mode + "_s"] = round(time.time() - t1, 1)
            del m
        res["schemes"][name] = row
        print(name, row, file=sys.stderr)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
