"""ar_attn_fwd_exact / ar_attn_bwd_exact against torch's attention over seeds, operand scales (small logits ... saturated softmax),
gradient scales and mask parameters, at the two minibatch shapes: any differing value is reported."""
import json, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
from auto_round_amd.exact_block import exact_attention_backward
res, bad = [], 0
def nd(a, b):
    a, b = a.contiguous(), b.contiguous()
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return int((a.view(it) != b.view(it)).sum())
cases = []
for seed in range(4):
    for qs, gs, valid_off in ((0.05, 1e-4, 1), (1.0, 0.02, 1), (4.0, 3.0, 700), (12.0, 0.5, 2047)):
        cases.append((seed, qs, gs, valid_off))
for (B, H, S, D, hk, scale) in ((8, 32, 2048, 128, 8, 128 ** -0.5), (8, 12, 2048, 64, 12, 1.0)):
    for (seed, qs, gs, valid_off) in cases:
        torch.manual_seed(seed)
        q = (torch.randn(B, S, H, D, device="cuda") * qs).to(torch.bfloat16).transpose(1, 2)
        k = (torch.randn(B, S, hk, D, device="cuda") * qs).to(torch.bfloat16).transpose(1, 2)
        v = (torch.randn(B, S, hk, D, device="cuda") * (1 + seed)).to(torch.bfloat16).transpose(1, 2)
        valid = S - valid_off
        idx = torch.arange(S, device="cuda")
        keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < valid)
        mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
        st = ops.mask_structure(mask, S)
        rep = H // hk
        ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
        ke = kl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else kl
        ve = vl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else vl
        ao = F.scaled_dot_product_attention(ql, ke, ve, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale).transpose(1, 2).contiguous()
        da = (torch.randn(B, S, H, D, device="cuda") * gs).to(torch.bfloat16)
        gq, gk, gv = torch.autograd.grad(ao, (ql, kl, vl), da)
        with torch.no_grad():
            mo, mlse = ops.attn_fwd_exact(q, k, v, st, scale)
            d = exact_attention_backward((q, k, v, mo, mlse, st), da, scale)
        rec = {"D": D, "seed": seed, "operand_std": qs, "grad_std": gs, "valid_len": valid, "out": nd(mo, ao.detach()),
               "dq": nd(d[0], gq), "dk": nd(d[1], gk), "dv": nd(d[2], gv), "nonfinite_ref": int((~torch.isfinite(gq.float())).sum())}
        bad += rec["out"] + rec["dq"] + rec["dk"] + rec["dv"]
        res.append(rec)
        print(json.dumps(rec), flush=True)
print("TOTAL differing values:", bad, "over", len(res), "cases")
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
json.dump({"cases": res, "total_differing": bad}, open(os.path.join(out, "attn_exact_fuzz.json"), "w"), indent=1)
