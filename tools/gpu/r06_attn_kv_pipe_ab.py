"""The hand-pipelined key-side backward kernel (k_xattn_bwd_kv) against the phase-by-phase default (the pipelined one: config bit 16): equal
results (dQ, dK, dV bit for bit, several mask lengths / sequence lengths) and the time of the whole backward call."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops, _lib
lib = _lib.load()
torch.manual_seed(0)
res = []
def tm(f, n=20):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
shapes = [(8, 32, 2048, 128, 8, 2047), (8, 32, 2048, 128, 8, 1300), (2, 32, 512, 128, 8, 511), (1, 16, 4096, 128, 16, 4000), (2, 8, 1152, 128, 2, 1100),
          (8, 64, 2048, 128, 8, 2047)]
if "--d64" in sys.argv:
    shapes += [(8, 12, 2048, 64, 12, 2047), (2, 12, 640, 64, 12, 600)]
for (B, H, S, D, hk, valid) in shapes:
    scale = D ** -0.5
    q = torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
    k = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
    idx = torch.arange(S, device="cuda")
    keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < valid)
    mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
    st = ops.mask_structure(mask, S)
    da = (torch.randn(B, S, H, D, device="cuda") * 0.02).to(torch.bfloat16)
    rec = {"shape": [B, H, S, D, hk], "valid_len": valid}
    with torch.no_grad():
        o, lse = ops.attn_fwd_exact(q, k, v, st, scale)
        outs = {}
        for name, cfg in (("phases", 0), ("pipelined", 16)):
            lib.ar_attn_exact_config(cfg)
            g = ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale)
            outs[name] = [t.clone() for t in g]
            rec[name + "_bwd_ms"] = round(tm(lambda: ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale)), 4)
        lib.ar_attn_exact_config(0)
    rec["differing"] = [int((a.view(torch.int16) != b.view(torch.int16)).sum()) for a, b in zip(outs["phases"], outs["pipelined"])]
    print(json.dumps(rec), flush=True)
    res.append(rec)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_kv_pipe_ab.json"), "w"), indent=1)
