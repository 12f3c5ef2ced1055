"""Which op under the fused OPT path is not bit-reproducible run to run?  (a) every kernel / library call of the path repeated on
the same inputs, outputs compared bit for bit; (b) the whole block tuned twice per configuration switch."""
import copy, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from auto_round_amd import ops

dev = torch.device("cuda:0")
B, S, H, hd, Hd, FF = 8, 2048, 12, 64, 768, 3072
T = B * S
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(torch.bfloat16)
out = {}


def repeat(name, fn, n=60):
    ref = fn()
    ref = [t.clone() for t in (ref if isinstance(ref, (tuple, list)) else [ref])]
    bad = 0
    for _ in range(n):
        r = fn()
        r = r if isinstance(r, (tuple, list)) else [r]
        if not all(torch.equal(a, b) for a, b in zip(ref, r)):
            bad += 1
    out[name] = bad
    print(name, "differing repeats:", bad, "of", n, flush=True)


qkv = rnd(T, 3 * Hd)
q, k, v = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
do = rnd(T, Hd, sc=0.1)
o, lse = ops.attn_fwd(q, k, v, B, S, H, hd, scale=0.125)
repeat("attn_fwd", lambda: ops.attn_fwd(q, k, v, B, S, H, hd, scale=0.125))
repeat("attn_bwd", lambda: ops.attn_bwd(q, k, v, o, lse, do, B, S, H, hd, scale=0.125))
x = rnd(T, Hd)
a = rnd(T, FF)
for nm, (n_out, n_in) in dict(qkv=(3 * Hd, Hd), o=(Hd, Hd), fc1=(FF, Hd), fc2=(Hd, FF)).items():
    W = rnd(n_out, n_in, sc=0.02)
    bias = rnd(n_out, sc=0.02)
    xin = a if n_in == FF else x
    dy = rnd(T, n_out, sc=0.01)
    repeat(f"linear_{nm}", lambda: F.linear(xin, W, bias))
    repeat(f"linear_nobias_{nm}", lambda: F.linear(xin, W))
    res = rnd(T, n_out)
    repeat(f"addmm_{nm}", lambda: torch.addmm(res, xin, W.t()))
    repeat(f"dx_{nm}", lambda: torch.mm(dy, W))
    Wt = W.t().contiguous()
    repeat(f"dx_tn_{nm}", lambda: torch.mm(dy, Wt.t()))
    dW = torch.empty(n_out, n_in, dtype=torch.bfloat16, device=dev)

    def dwf():
        assert ops.gemm_dw(dy, xin, dW, accumulate=False)
        return dW
    repeat(f"gemm_dw_{nm}", dwf)
    repeat(f"lib_dw_{nm}", lambda: torch.mm(dy.t(), xin))
w = torch.ones(Hd, dtype=torch.bfloat16, device=dev)
bz = torch.zeros(Hd, dtype=torch.bfloat16, device=dev)
repeat("layernorm_fwd", lambda: ops.layernorm_fwd(x, w, bz, 1e-5, want_stats=True))
json.dump(out, open(os.path.join(os.environ.get("OUT", "."), "det_ops.json"), "w"), indent=1)

# ---- (b) the block, twice per configuration
import transformers
from auto_round_amd.autoround import loss_mask_ids
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
from auto_round_amd.schemes import apply_scheme, resolve_scheme
from auto_round_amd.testing import t3_fixture as fx

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
for name, kw in dict(default={}, no_graph=dict(hip_graph=False), no_attn_bwd=dict(flash_attention_bwd=False, hip_graph=False),
                     no_flash=dict(flash_attention=False, hip_graph=False), no_mfma_dw=dict(mfma_dw_gemm=False, hip_graph=False),
                     no_tn_dx=dict(tn_dx_gemm=False, hip_graph=False)).items():
    traces = []
    for _ in range(3):
        blk = copy.deepcopy(block)
        cfg = dict(iters=30, batch_size=8, bits=4, fused_block=True, mfma_dw_gemm=True)
        cfg.update(kw)
        qz = SignRoundQuantizer(SignRoundConfig(**cfg), device=dev)
        transformers.set_seed(42)
        qz.quantize_block(blk, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=ids)
        torch.cuda.synchronize()
        traces.append(list(qz.last_stats["loss_trace"]))
    first = [next((i for i, (p, r) in enumerate(zip(traces[0], t)) if p != r), None) for t in traces[1:]]
    res[name] = first
    print(name, "first differing iteration vs run 0:", first, flush=True)
json.dump(res, open(os.path.join(os.environ.get("OUT", "."), "det_block.json"), "w"), indent=1)
