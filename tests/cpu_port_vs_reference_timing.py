"""Build container only: is the CPU baseline `bench.py` reports (kind "port": oracle/torch_ref timed on the host) representative
of the REAL reference's CPU path?  Times the reference's own `AutoRound(...).quantize()` tuning loop and the restated flow
(tests/pipeline_flow.py + oracle/torch_ref) on the same OPT-125M-shaped 2-block model, same calibration data, same iterations, same
thread count, and prints seconds per tuning iteration for both.  -> profiles/archive/r01_cpu_port_vs_reference_timing.json"""
import copy
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.dont_write_bytecode = True
for p in (os.path.join(ROOT, "oracle", "ref_shim"), "/root/reference", os.path.join(ROOT, "tests"), ROOT):     # (lives under tests/: only tests may use oracle/)
    sys.path.insert(0, p)

from transformers import OPTConfig, OPTForCausalLM  # noqa: E402

from auto_round import AutoRound  # noqa: E402
from auto_round.algorithms.quantization.sign_round import quantizer as RQ  # noqa: E402
from auto_round_amd.schemes import apply_scheme, resolve_scheme  # noqa: E402
from oracle import torch_ref as tr  # noqa: E402
from pipeline_flow import run_flow  # noqa: E402
from test_pipeline_vs_reference import _Loader, _StubTokenizer  # noqa: E402

iters, nsamples, seqlen, bs = (int(os.environ.get("ITERS", 8)), int(os.environ.get("NSAMPLES", 16)), int(os.environ.get("SEQLEN", 512)), 8)
layers = int(os.environ.get("LAYERS", 2))          # ITERS=200 NSAMPLES=128 SEQLEN=2048 LAYERS=1 = one block of BASELINE configs[0]
torch.manual_seed(0)
cfg = OPTConfig(hidden_size=768, ffn_dim=3072, num_attention_heads=12, num_hidden_layers=layers, vocab_size=50272, max_position_embeddings=2048)
cfg._attn_implementation = "sdpa"
base = OPTForCausalLM(cfg).to(torch.bfloat16).eval()
tokens = torch.randint(0, 50272, (nsamples, seqlen), generator=torch.Generator().manual_seed(1))
os.makedirs("/tmp/cpu_timing", exist_ok=True)
os.chdir("/tmp/cpu_timing")

# --- the reference: time only its per-block tuning loop (quantize_block), like the port below
spent = {"ref": 0.0, "port": 0.0}
orig_qb = RQ.SignRoundQuantizer.quantize_block


def timed_qb(self, *a, **k):
    t0 = time.perf_counter()
    try:
        return orig_qb(self, *a, **k)
    finally:
        spent["ref"] += time.perf_counter() - t0


RQ.SignRoundQuantizer.quantize_block = timed_qb
ar = AutoRound(copy.deepcopy(base), tokenizer=_StubTokenizer(), iters=iters, nsamples=nsamples, seqlen=seqlen, dataset=_Loader(tokens),
               device_map="cpu", batch_size=bs, enable_torch_compile=False, scheme="W4A16")
ar.quantize()
RQ.SignRoundQuantizer.quantize_block = orig_qb

# --- the port
m = copy.deepcopy(base)
for p in m.parameters():
    p.requires_grad_(False)
sch = resolve_scheme("W4A16")
blocks = list(m.model.decoder.layers)
for b in blocks:
    apply_scheme(b, sch)
orig_tb = tr.tune_block


def timed_tb(*a, **k):
    t0 = time.perf_counter()
    try:
        return orig_tb(*a, **k)
    finally:
        spent["port"] += time.perf_counter() - t0


tr.tune_block = timed_tb
run_flow(m, blocks, tokens, sch, iters=iters, bs=bs, reference_mask=True)
out = dict(model=f"OPT-125M-shaped, {layers} block(s) (hidden 768, ffn 3072, 12 heads), W4G128 sym", nsamples=nsamples, seqlen=seqlen, batch_size=bs,
           iters=iters, threads=torch.get_num_threads(),
           reference_tuning_s_per_iter=spent["ref"] / (layers * iters), port_tuning_s_per_iter=spent["port"] / (layers * iters),
           reference_tuning_s_per_block=spent["ref"] / layers, port_tuning_s_per_block=spent["port"] / layers,
           port_over_reference=spent["port"] / spent["ref"])
print(json.dumps(out))
