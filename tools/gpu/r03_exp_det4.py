"""Flake rates of the LIBRARY GEMM forms under the fused OPT block (OPT-125M shapes, T = 16384 tokens, bf16): each call repeated on
the same operands, results compared bit for bit -- default settings, then under torch.use_deterministic_algorithms(True)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.nn.functional as F
from auto_round_amd import ops

dev = torch.device("cuda:0")
T, Hd, FF = 16384, 768, 3072
g = torch.Generator(device=dev).manual_seed(1)
rnd = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).to(torch.bfloat16)
N = int(os.environ.get("REPEATS", "600"))


def flakes(fn):
    ref = fn().clone()
    bad = 0
    for _ in range(N):
        if not torch.equal(ref, fn()):
            bad += 1
    return bad


x, a = rnd(T, Hd), rnd(T, FF)
forms = {}
for nm, (n_out, n_in) in dict(qkv=(3 * Hd, Hd), o=(Hd, Hd), fc1=(FF, Hd), fc2=(Hd, FF)).items():
    W, bias = rnd(n_out, n_in, sc=0.02), rnd(n_out, sc=0.02)
    xin = a if n_in == FF else x
    dy, res = rnd(T, n_out, sc=0.01), rnd(T, n_out)
    forms[f"fwd_linear_bias_{nm}"] = (lambda xin=xin, W=W, bias=bias: F.linear(xin, W, bias))
    forms[f"fwd_addmm_inplace_{nm}"] = (lambda xin=xin, W=W, res=res: res.clone().addmm_(xin, W.t()))
    forms[f"dx_nn_{nm}"] = (lambda dy=dy, W=W: torch.mm(dy, W))
    forms[f"dw_lib_{nm}"] = (lambda dy=dy, xin=xin: torch.mm(dy.t(), xin))
out = {}
for mode in ("default", "deterministic_algorithms"):
    if mode != "default":
        torch.use_deterministic_algorithms(True, warn_only=True)
    out[mode] = {k: flakes(f) for k, f in forms.items()}
    print(mode, json.dumps(out[mode]), flush=True)
torch.use_deterministic_algorithms(False)
# first-party kernels of the same path, same count
w, bz = torch.ones(Hd, dtype=torch.bfloat16, device=dev), torch.zeros(Hd, dtype=torch.bfloat16, device=dev)
yln, mean, rstd = ops.layernorm_fwd(x, w, bz, 1e-5, want_stats=True)
dyl = rnd(T, Hd, sc=0.01)
qkv = rnd(T, 3 * Hd)
q, k, v = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
o, lse = ops.attn_fwd(q, k, v, 8, 2048, 12, 64, scale=0.125)
do = rnd(T, Hd, sc=0.1)
dW = torch.empty(FF, Hd, dtype=torch.bfloat16, device=dev)
dyf = rnd(T, FF, sc=0.01)


def dwf():
    ops.gemm_dw(dyf, x, dW, accumulate=False)
    return dW


mine = dict(layernorm_fwd=lambda: ops.layernorm_fwd(x, w, bz, 1e-5, want_stats=True)[0],
            layernorm_bwd=lambda: ops.layernorm_bwd(dyl, x, w, mean, rstd, dres=dyl),
            attn_fwd=lambda: ops.attn_fwd(q, k, v, 8, 2048, 12, 64, scale=0.125)[0],
            attn_bwd_dq=lambda: ops.attn_bwd(q, k, v, o, lse, do, 8, 2048, 12, 64, scale=0.125)[0],
            attn_bwd_dk=lambda: ops.attn_bwd(q, k, v, o, lse, do, 8, 2048, 12, 64, scale=0.125)[1],
            gemm_dw_fc1=dwf)
out["first_party"] = {k_: flakes(f) for k_, f in mine.items()}
print("first_party", json.dumps(out["first_party"]), flush=True)
out["repeats"] = N
json.dump(out, open(os.path.join(os.environ.get("OUT", "."), "det_library_gemm_flakes.json"), "w"), indent=1)
