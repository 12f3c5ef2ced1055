"""Which CLASS of ops makes two runs of the fused OPT block part?  All ops synchronised -> bit-identical (r03_exp_det5.py); nothing
synchronised -> they part.  Here only one class at a time is bracketed by torch.cuda.synchronize(): the library GEMMs, the first-party
MFMA kernels (weight-gradient GEMM, attention forward / backward), or the quant kernels."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
import transformers
from auto_round_amd import ops
from auto_round_amd.autoround import loss_mask_ids
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
from auto_round_amd.schemes import apply_scheme, resolve_scheme
from auto_round_amd.testing import t3_fixture as fx

dev = torch.device("cuda:0")
ACTIVE = set()


def bracket(cls, fn):
    def w(*a, **k):
        if cls in ACTIVE:
            torch.cuda.synchronize()
        r = fn(*a, **k)
        if cls in ACTIVE:
            torch.cuda.synchronize()
        return r
    return w


torch.mm = bracket("lib", torch.mm)
torch.addmm = bracket("lib", torch.addmm)
torch.Tensor.addmm_ = bracket("lib", torch.Tensor.addmm_)
F.linear = bracket("lib", F.linear)
torch.nn.functional.linear = F.linear
for nm in ("gemm_dw", "attn_fwd", "attn_bwd"):
    setattr(ops, nm, bracket("mfma", getattr(ops, nm)))
for nm in ("qdq_int_fwd", "qdq_int_bwd_sgd_", "qdq_int_bwd", "mse_loss_fwd_bwd", "gather_rows", "layernorm_fwd", "layernorm_bwd", "best_loss_update"):
    if hasattr(ops, nm):
        setattr(ops, nm, bracket("quant", getattr(ops, nm)))

# heat: a burst of GEMMs like the probe in which every run parted at iteration 1
a = torch.randn(16384, 3072, device=dev, dtype=torch.bfloat16)
b = torch.randn(3072, 3072, device=dev, dtype=torch.bfloat16)
for _ in range(int(os.environ.get("BURN", "3000"))):
    torch.mm(a, b)
torch.cuda.synchronize()

model = fx.build_model("opt125m").to(dev)
for p in model.parameters():
    p.requires_grad_(False)
tokens = fx.calib_tokens("opt125m", 128, 2048)
block = fx.decoder_blocks(model)[0]
apply_scheme(block, resolve_scheme("W4A16"))
x0, others = fx.capture_block_inputs(model, block, tokens, dev)
ids = loss_mask_ids(tokens, None)
y = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, fused_block=False), device=dev).calibrate_block(block, x0, others)
res = {}
for name, classes in (("none", ()), ("lib", ("lib",)), ("mfma", ("mfma",)), ("quant", ("quant",)), ("lib+mfma", ("lib", "mfma")), ("none_again", ())):
    ACTIVE.clear()
    ACTIVE.update(classes)
    traces = []
    for _ in range(4):
        blk = copy.deepcopy(block)
        qz = SignRoundQuantizer(SignRoundConfig(iters=20, batch_size=8, bits=4, fused_block=True, mfma_dw_gemm=True, hip_graph=False), device=dev)
        transformers.set_seed(42)
        qz.quantize_block(blk, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=ids)
        torch.cuda.synchronize()
        traces.append([float(v) for v in qz.last_stats["loss_trace"]])
    first = [next((i for i, (p, r) in enumerate(zip(traces[0], t)) if p != r), None) for t in traces[1:]]
    res[name] = first
    print(name, "first differing iteration of runs 1..3 vs run 0:", first, flush=True)
json.dump(res, open(os.path.join(os.environ.get("OUT", "."), "det_by_class.json"), "w"), indent=1)
