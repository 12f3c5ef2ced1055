"""Which attention arithmetic did the REAL reference's process run when it produced the Mixtral fixtures' targets?  The fixtures keep the
sha256 of the targets (y_sha: the fp block's outputs the reference tuned against); the reference-free flow's differ in their last bits
(identical q / k / v, other attention output: profiles/r05_t3_mixtral_forward_compare_*.json).  This probe recomputes the targets with the
block's attention swapped for every candidate this stack offers -- the library's SDPA (inference and training-mode forward), its math
backend, transformers' eager attention, and the restated AOTriton forward at each key-block size (another tuned configuration of the same
kernel) -- and, last, with the rotary cos / sin tables made on the CPU instead of on the GPU, and compares each result's sha256 with the
fixture's.  Result (profiles/r06_mixtral_targets_probe.json): only the CPU-made tables reproduce it -- the reference captures the first
block's inputs with the model on the CPU (calibration/llm.py:74-90)."""
import json, os, sys, hashlib
import numpy as np
import torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd.testing import t3_fixture as fx
from auto_round_amd import attention as att
from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
from auto_round_amd.schemes import apply_scheme, resolve_scheme

root = os.environ.get("GRAFT_REPO_ROOT", ".")
path = os.path.join(root, "tests", "golden", "t3s_mixtral8x7b_mxfp4_100.npz")
m = json.loads(str(np.load(path, allow_pickle=False)["meta"]))
arch = m["arch"]
dev = torch.device("cuda:0")
torch.use_deterministic_algorithms(True, warn_only=True)          # the reference's process-global mode (compressors/base.py:339-351)
model = fx.build_model(arch).to(dev)
for p in model.parameters():
    p.requires_grad_(False)
tokens = fx.calib_tokens(arch, m["nsamples"], m["seqlen"])
block = fx.decoder_blocks(model)[0]
from auto_round_amd.moe_unfuse import unfuse_moe_experts
unfuse_moe_experts(model)
kw = dict(m.get("scheme_kw") or {}); kw.pop("enable_alg_ext", None); kw.pop("lr", None); kw.pop("minmax_lr", None)
sch = resolve_scheme(m["scheme"], **kw)
apply_scheme(block, sch)
x0, others = fx.capture_block_inputs(model, block, tokens, dev)
print("x_sha identical:", fx.sha(x0) == m["x_sha"], "attn implementation:", model.config._attn_implementation, flush=True)
res = {"fixture": os.path.basename(path), "inputs_identical": fx.sha(x0) == m["x_sha"], "variants": {}}
base = None

def targets(name, reproducible, impl, kb=0, math=False):
    global base
    cfg = SignRoundConfig(iters=2, batch_size=m["batch_size"], bits=sch["bits"], sdpa_backend="auto", fused_block=False,
                          materialise_shared_rows=True, reproducible_attention_forward=reproducible)
    q = SignRoundQuantizer(cfg, device=dev)
    old = model.config._attn_implementation
    model.config._attn_implementation = impl
    att.exact_state.update(key_block=kb, verify=False, calls=0, fallbacks=0)
    try:
        ctx = torch.nn.attention.sdpa_kernel([torch.nn.attention.SDPBackend.MATH]) if math else __import__("contextlib").nullcontext()
        with torch.cuda.device(dev), ctx:
            y = q.calibrate_block(block, x0, others)
    finally:
        model.config._attn_implementation = old
    rec = {"targets_identical_to_fixture": fx.sha(y) == m["y_sha"]}
    if impl == att.EXACT_NAME:
        rec.update(exact_calls=att.exact_state["calls"], exact_fallbacks=att.exact_state["fallbacks"])
    if base is None:
        base = y
    else:
        d = (y.view(torch.int16) != base.view(torch.int16))
        rec["values_differing_from_library_form"] = float(d.float().mean())
    res["variants"][name] = rec
    print(name, json.dumps(rec), flush=True)
    del y

targets("library sdpa, training-mode forward (the product's)", True, "sdpa")
targets("library sdpa, inference-mode forward", False, "sdpa")
targets("library sdpa, math backend", False, "sdpa", math=True)
targets("transformers eager attention", False, "eager")
att.register_exact_sdpa()
for kb in (64, 32, 16):
    targets(f"restated AOTriton forward, key block {kb}", False, att.EXACT_NAME, kb=kb)
# ---- the rotary tables: cos / sin made on the CPU (fp32 cos / sin of another libm, then rounded to bf16) instead of on the GPU
import copy
pe = others.get("position_embeddings")
if pe is not None:
    rot = copy.deepcopy(model.model.rotary_emb).to("cpu")
    pos = others.get("position_ids")
    pos_cpu = (pos if pos is not None else torch.arange(m["seqlen"])[None]).to("cpu")
    dummy = torch.zeros(1, m["seqlen"], 8, dtype=pe[0].dtype)
    cos_c, sin_c = rot(dummy, pos_cpu)
    dc = int((cos_c.to(dev).view(torch.int16) != pe[0].view(torch.int16)).sum()); ds = int((sin_c.to(dev).view(torch.int16) != pe[1].view(torch.int16)).sum())
    res["rotary_tables_cpu_vs_gpu"] = {"cos_differing": dc, "sin_differing": ds, "of": int(pe[0].numel()), "dtype": str(pe[0].dtype), "shape": list(pe[0].shape)}
    print("rotary tables made on the CPU vs on the GPU:", res["rotary_tables_cpu_vs_gpu"], flush=True)
    keep = others["position_embeddings"]
    others["position_embeddings"] = (cos_c.to(dev).reshape(pe[0].shape).contiguous(), sin_c.to(dev).reshape(pe[1].shape).contiguous())
    targets("library sdpa, rotary tables made on the CPU", True, "sdpa")
    others["position_embeddings"] = keep
out = os.path.join(root, "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "mixtral_targets_probe.json"), "w"), indent=1)
