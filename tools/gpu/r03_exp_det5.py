"""Where do two identical runs of the fused OPT block part?  Every tensor the block saves in its forward, its output, the incoming
gradient and the weight-gradient arena after its backward are checksummed (exact integer sums of the bit patterns) call by call;
the first differing entry between run A and run B names the op."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import transformers
from auto_round_amd import fused_block as fbm
from auto_round_amd.autoround import loss_mask_ids
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
from auto_round_amd.schemes import apply_scheme, resolve_scheme
from auto_round_amd.testing import t3_fixture as fx

dev = torch.device("cuda:0")
LOG = []


def cs(t):
    if t is None:
        return None
    t = t.detach().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        v = t.view(torch.int16)
    elif t.dtype == torch.float32:
        v = t.view(torch.int32)
    else:
        v = t
    return int(v.to(torch.int64).sum().item()) ^ int((v.to(torch.int64) * 31 % 1000003).sum().item())


fwd0, bwd0 = fbm.FusedOPTBlock._forward_impl, fbm.FusedOPTBlock._backward_impl


def fwd(self, x, others, ctx):
    LOG.append(("in:x", cs(x)))
    LOG.append(("in:Wqkv", cs(self.Wqkv)))
    LOG.append(("in:W1", cs(self.W1)))
    y = fwd0(self, x, others, ctx)
    if ctx is not None:
        for k_, v in ctx.saved.items():
            if isinstance(v, torch.Tensor):
                LOG.append(("fwd:" + k_, cs(v)))
            elif k_ == "leaves" and isinstance(v, tuple) and isinstance(v[0], str):
                for nm, t in zip(("q", "k", "v", "out", "lse"), v[1:]):
                    LOG.append(("fwd:leaf_" + nm, cs(t)))
    LOG.append(("fwd:y", cs(y)))
    return y


def bwd(self, ctx, dy):
    LOG.append(("bwd:dy", cs(dy)))
    r = bwd0(self, ctx, dy)
    for nm in ("dWqkv", "dWo", "dW1", "dW2"):
        LOG.append(("bwd:" + nm, cs(getattr(self, nm))))
    return r


fbm.FusedOPTBlock._forward_impl, fbm.FusedOPTBlock._backward_impl = fwd, bwd
model = fx.build_model("opt125m").to(dev)
for p in model.parameters():
    p.requires_grad_(False)
tokens = fx.calib_tokens("opt125m", 128, 2048)
block = fx.decoder_blocks(model)[0]
apply_scheme(block, resolve_scheme("W4A16"))
x0, others = fx.capture_block_inputs(model, block, tokens, dev)
ids = loss_mask_ids(tokens, None)
y = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, fused_block=False), device=dev).calibrate_block(block, x0, others)
logs = []
for run in range(int(os.environ.get("RUNS", "4"))):
    LOG.clear()
    blk = copy.deepcopy(block)
    qz = SignRoundQuantizer(SignRoundConfig(iters=10, batch_size=8, bits=4, fused_block=True, mfma_dw_gemm=True, hip_graph=False), device=dev)
    transformers.set_seed(42)
    qz.quantize_block(blk, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=ids)
    torch.cuda.synchronize()
    logs.append(list(LOG))
    print("run", run, "entries", len(LOG), "trace", ["%.9e" % float(v) for v in qz.last_stats["loss_trace"][:4]], flush=True)
rep = []
for r in range(1, len(logs)):
    a, b = logs[0], logs[r]
    first = next((i for i, (p, q) in enumerate(zip(a, b)) if p != q), None)
    if first is None:
        rep.append(dict(run=r, first=None))
        print("run", r, "identical to run 0 (", len(a), "entries )")
    else:
        ctx_ = [a[j][0] for j in range(max(0, first - 3), min(len(a), first + 4))]
        diffs = [a[j][0] for j in range(first, min(len(a), first + 40)) if a[j] != b[j]]
        rep.append(dict(run=r, first_index=first, first_name=a[first][0], around=ctx_, next_differing=diffs[:12]))
        print("run", r, "first difference at entry", first, a[first][0], "| then:", diffs[:12], flush=True)
json.dump(rep, open(os.path.join(os.environ.get("OUT", "."), "det_first_difference.json"), "w"), indent=1)
