"""Where do ar_attn_fwd_exact / ar_attn_bwd_exact equal torch's attention?  The library picks its kernel configuration (tile sizes) by
problem shape; the kernels restate the configurations of the two tuning-minibatch shapes.  A grid of other shapes: differing values of
the output / log-sum-exp (forward) and of dQ / dK / dV (backward)."""
import json, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
from auto_round_amd.exact_block import exact_attention_backward
torch.manual_seed(0)
res = []
def nd(a, b):
    a, b = a.contiguous(), b.contiguous()
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return int((a.view(it) != b.view(it)).sum())
for (H, D, hk) in ((32, 128, 8), (12, 64, 12), (16, 64, 16), (8, 128, 8), (64, 128, 8)):
    for S in (256, 512, 1024, 2048, 4096):
        for B in (1, 4, 8):
            if B * H * S * S > 8 * 64 * 2048 * 2048:
                continue
            scale = 1.0 if D == 64 else D ** -0.5
            q = (torch.randn(B, S, H, D, device="cuda") * (0.35 if D == 64 else 1.0)).to(torch.bfloat16).transpose(1, 2)
            k = (torch.randn(B, S, hk, D, device="cuda") * (0.35 if D == 64 else 1.0)).to(torch.bfloat16).transpose(1, 2)
            v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
            idx = torch.arange(S, device="cuda")
            keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
            mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
            st = ops.mask_structure(mask, S)
            rep = H // hk
            ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
            ke = kl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else kl
            ve = vl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else vl
            ao = F.scaled_dot_product_attention(ql, ke, ve, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale).transpose(1, 2).contiguous()
            da = (torch.randn(B, S, H, D, device="cuda") * 0.02).to(torch.bfloat16)
            gq, gk, gv = torch.autograd.grad(ao, (ql, kl, vl), da)
            with torch.no_grad():
                got = ops.attn_fwd_exact(q, k, v, st, scale)
                rec = {"B": B, "H": H, "S": S, "D": D, "hk": hk}
                if got is None:
                    rec["fwd"] = "refused"
                else:
                    mo, mlse = got
                    rec["out_differ"] = nd(mo, ao.detach())
                    # backward on torch's own forward results (so that it is judged by itself)
                    lse_ref = torch.ops.aten._scaled_dot_product_efficient_attention(q, ke.detach(), ve.detach(), mask.expand(B, H, S, S), True, 0.0, False, scale=scale)[1]
                    rec["lse_differ"] = nd(mlse, lse_ref[..., :S])
                    try:
                        d = exact_attention_backward((q, k, v, ao.detach(), lse_ref[..., :S].contiguous(), st), da, scale)
                        rec["dq_differ"], rec["dk_differ"], rec["dv_differ"] = nd(d[0], gq), nd(d[1], gk), nd(d[2], gv)
                    except Exception as e:
                        rec["bwd"] = repr(e)[:80]
            print(json.dumps(rec), flush=True)
            res.append(rec)
            del q, k, v, mask, ao, gq, gk, gv
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_exact_shape_grid.json"), "w"), indent=1)
