"""Timing ablations of the key-side backward kernel (library built with EXTRA=-DAR_XATTN_ABLATIONS; results are WRONG by construction):
which part of the kernel the time goes to.  Prints the time of the whole backward call minus the dQ + preprocess kernels."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops, _lib
lib = _lib.load()
torch.manual_seed(0)
B, H, S, D, hk = 8, 32, 2048, 128, 8
scale = D ** -0.5
q = torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
k = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
idx = torch.arange(S, device="cuda")
keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
st = ops.mask_structure(mask, S)
da = (torch.randn(B, S, H, D, device="cuda") * 0.02).to(torch.bfloat16)
def tm(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
names = {0: "full", 1: "no softmax VALU", 2: "no LDS reads", 3: "no VALU, no LDS reads", 4: "no DMA", 12: "no DMA, no barrier", 15: "only MFMAs (+ waits)",
         64: "DMA issued, never waited for", 72: "DMA never waited for, no barrier", 66: "no LDS reads, DMA not waited", 67: "no VALU, no LDS reads, DMA not waited", 16: "no score MFMAs", 32: "no accumulating MFMAs", 48: "no MFMAs", 63: "nothing"}
res = {}
with torch.no_grad():
    o, lse = ops.attn_fwd_exact(q, k, v, st, scale)
    for abl, name in names.items():
        lib.ar_attn_exact_config(abl << 8)
        res[name] = round(tm(lambda: ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale)), 4)
        print(f"{name:28s} {res[name]:.3f} ms per backward call", flush=True)
    lib.ar_attn_exact_config(0)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_kv_ablation.json"), "w"), indent=1)
