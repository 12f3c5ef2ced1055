"""GPU box: quantise a tiny random Llama with the standalone front door and write the checkpoint + the tuned model's logits
(tests/golden/tiny_ckpt_*): the CPU test then loads that checkpoint through the REFERENCE's own inference loader."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_autoround import tiny_llama  # noqa: E402

from auto_round_amd.autoround import AutoRound  # noqa: E402

out_root, only = sys.argv[1], sys.argv[2:]          # optional tags: regenerate just those fixtures
for tag, kw in (("w4g32_sym", dict(scheme="W4A16", group_size=32)), ("w2g32_asym", dict(scheme="W2A16G32", sym=False)),
                ("w2g32_sym_algext", dict(scheme="W2A16G32", enable_alg_ext=True)), ("w3g32_sym", dict(scheme="W3A16", group_size=32)),
                ("mxfp4", dict(scheme="MXFP4")), ("nvfp4", dict(scheme="NVFP4", enable_alg_ext=True)),
                ("w4g32_sym_fmt_gptq", dict(scheme="W4A16", group_size=32, format="auto_gptq")),
                ("w4g32_asym_fmt_awq", dict(scheme="W4A16", group_size=32, sym=False, format="auto_awq")),
                ("nvfp4_fmt_llmc", dict(scheme="NVFP4", format="llm_compressor")),
                ("mxfp4_fmt_llmc", dict(scheme="MXFP4", format="llm_compressor")),
                ("gpt2_w4g32_sym", dict(scheme="W4A16", group_size=32, arch="gpt2"))):
    if only and tag not in only:
        continue
    kw = dict(kw)
    fmt = kw.pop("format", "auto_round")
    arch = kw.pop("arch", "llama")
    if arch == "gpt2":       # Conv1D projections (weights stored [in, out]), fused qkv, blocks under transformer.h
        from transformers import GPT2Config, GPT2LMHeadModel

        torch.manual_seed(3)
        gcfg = GPT2Config(n_embd=128, n_head=4, n_layer=2, n_inner=256, vocab_size=64, n_positions=64)
        gcfg._attn_implementation = "sdpa"
        model = GPT2LMHeadModel(gcfg).to(torch.bfloat16)
        prefix, names = "transformer.h.0.", ("attn.c_attn", "mlp.c_proj")
    else:
        model = tiny_llama(seed=3, vocab=64)
        prefix, names = "model.layers.0.", ("self_attn.q_proj", "mlp.down_proj")
    g = torch.Generator().manual_seed(1)
    tokens = torch.randint(0, 64, (8, 32), generator=g)
    ar = AutoRound(model, None, iters=6, nsamples=8, seqlen=32, batch_size=4, dataset=tokens, **kw)
    out = os.path.join(out_root, f"tiny_ckpt_{tag}")
    qmodel, _ = ar.quantize_and_save(out, format=fmt)
    with torch.no_grad():
        logits = qmodel(input_ids=tokens[:2].cuda()).logits.float().cpu().numpy()
    def lin(n):          # A4 schemes leave the activation-quant shell around the layer
        m = model.get_submodule(f"{prefix}{n}")
        return getattr(m, "orig_layer", m)

    baked = {f"W_{n.replace('.', '_')}": lin(n).weight.detach().cpu().view(torch.int16).numpy()
             for n in names}      # bf16 bit patterns of two tuned weights (as stored: Conv1D keeps [in, out])
    np.savez_compressed(os.path.join(out, "expected.npz"), tokens=tokens[:2].numpy(), logits=logits, prefix=np.array(prefix),
                        names=np.array(names), **baked)
    print(tag, "ok", os.listdir(out))
