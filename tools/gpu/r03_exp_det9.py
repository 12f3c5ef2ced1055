"""The forward tensor that flakes first: frozen-state loop of r03_exp_det7.py (fused OPT block, same minibatch, no sign-SGD step),
checksums of every tensor the forward saves taken ON THE GPU (no host synchronisation inside the loop), compared at the end."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import transformers
from auto_round_amd import fused_block as fbm
from auto_round_amd import ops
from auto_round_amd.autoround import loss_mask_ids
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
from auto_round_amd.schemes import apply_scheme, resolve_scheme
from auto_round_amd.testing import t3_fixture as fx

dev = torch.device("cuda:0")
ITERS = int(os.environ.get("ITERS", "1500"))
NAMES, ROWS = [], []


def cs(t):
    t = t.detach()
    if not t.is_contiguous():
        t = t.contiguous()
    v = (t.view(torch.int16) if t.element_size() == 2 else t.view(torch.int32)).to(torch.int32)
    return v.sum(dtype=torch.int64) + (v * 7 % 8191).sum(dtype=torch.int64)


ops.qdq_int_bwd_sgd_ = lambda *a, **k: None
fwd0 = fbm.FusedOPTBlock._forward_impl


def fwd(self, x, others, ctx):
    row, names = [cs(x), cs(self.Wqkv), cs(self.Wo), cs(self.W1), cs(self.W2)], ["x", "Wqkv", "Wo", "W1", "W2"]
    y = fwd0(self, x, others, ctx)
    if ctx is not None:
        for k_ in ("h1", "leaves", "attn2d", "x2", "mean2", "rstd2", "h2", "a"):
            v = ctx.saved[k_]
            if k_ == "leaves":
                for nm, t in zip(("q", "k", "v", "attn_out", "lse"), v[1:]):
                    row.append(cs(t)); names.append(nm)
            else:
                row.append(cs(v)); names.append(k_)
        row.append(cs(y)); names.append("y")
        ROWS.append(torch.stack(row))
        if not NAMES:
            NAMES.extend(names)
    return y


fbm.FusedOPTBlock._forward_impl = fwd
model = fx.build_model("opt125m").to(dev)
for p in model.parameters():
    p.requires_grad_(False)
tokens = fx.calib_tokens("opt125m", 128, 2048)
block = fx.decoder_blocks(model)[0]
apply_scheme(block, resolve_scheme("W4A16"))
x0, others = fx.capture_block_inputs(model, block, tokens, dev)
ids = loss_mask_ids(tokens, None)
y = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, fused_block=False), device=dev).calibrate_block(block, x0, others)
sched = [list(range(8))] * ITERS
blk = copy.deepcopy(block)
qz = SignRoundQuantizer(SignRoundConfig(iters=ITERS, batch_size=8, bits=4, fused_block=True, mfma_dw_gemm=True, hip_graph=False, not_use_best_mse=True), device=dev)
transformers.set_seed(42)
qz.quantize_block(blk, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=ids, index_schedule=sched)
torch.cuda.synchronize()
S = torch.stack(ROWS).cpu()
bad = S != S[0]
rep = dict(iterations=int(S.shape[0]), names=NAMES, flaky_per_tensor={n: int(bad[:, i].sum()) for i, n in enumerate(NAMES)},
           first_flaky_tensor_per_flaky_iteration=[(int(i), NAMES[int(torch.nonzero(bad[i]).flatten()[0])]) for i in torch.nonzero(bad.any(1)).flatten()[:20]])
print(json.dumps(rep), flush=True)
json.dump(rep, open(os.path.join(os.environ.get("OUT", "."), "det_first_flaky_forward_tensor.json"), "w"), indent=1)
