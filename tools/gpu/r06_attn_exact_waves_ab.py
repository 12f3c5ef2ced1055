"""A/B of the launch forms of ar_attn_fwd_exact / ar_attn_bwd_exact (workgroups of 4 or 8 waves) at the two minibatch shapes; results
are checked equal between the forms."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops, _lib
lib = _lib.load()
torch.manual_seed(0)
res = []
def tm(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for (B, H, S, D, hk, scale, std) in ((8, 32, 2048, 128, 8, 128 ** -0.5, 1.0), (8, 12, 2048, 64, 12, 1.0, 0.35), (8, 64, 2048, 128, 8, 128 ** -0.5, 1.0)):
    q = (torch.randn(B, S, H, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
    k = (torch.randn(B, S, hk, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
    idx = torch.arange(S, device="cuda")
    keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
    mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
    st = ops.mask_structure(mask, S)
    da = (torch.randn(B, S, H, D, device="cuda") * 0.02).to(torch.bfloat16)
    rec = {"shape": [B, H, S, D, hk]}
    base = None
    with torch.no_grad():
        for name, cfg in (("waves8", 2 | (2 << 2)), ("waves4", 1 | (1 << 2)), ("default", 0), ("fused_key_side", 3 << 2)):
            lib.ar_attn_exact_config(cfg)
            o, lse = ops.attn_fwd_exact(q, k, v, st, scale)
            g = ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale)
            outs = [o, lse] + [t.clone() for t in g]
            if base is None:
                base = outs
            else:
                rec["forms_equal"] = all(torch.equal(a, b) for a, b in zip(base, outs))
            rec[name + "_fwd_ms"] = tm(lambda: ops.attn_fwd_exact(q, k, v, st, scale))
            rec[name + "_bwd_ms"] = tm(lambda: ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale))
        lib.ar_attn_exact_config(0)
    print(json.dumps(rec), flush=True)
    res.append(rec)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_exact_waves_ab.json"), "w"), indent=1)
