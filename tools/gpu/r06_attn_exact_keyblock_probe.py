"""Which key block (keys per online-softmax step) reproduces torch's attention FORWARD at the shapes where the tuning minibatch's
configuration does not (S <= 512; S = 4096 at head size 64)?  ar_attn_fwd_exact with key_block 16 / 32 / 64 against torch."""
import json, os, sys, torch
import torch.nn.functional as F
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
torch.manual_seed(0)
res = []
def nd(a, b):
    a, b = a.contiguous(), b.contiguous()
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return int((a.view(it) != b.view(it)).sum())
for (H, D, hk) in ((32, 128, 8), (12, 64, 12)):
    for S in (128, 256, 384, 512, 768, 1024, 2048, 4096):
        B = 4
        scale = 1.0 if D == 64 else D ** -0.5
        q = (torch.randn(B, S, H, D, device="cuda") * (0.35 if D == 64 else 1.0)).to(torch.bfloat16).transpose(1, 2)
        k = (torch.randn(B, S, hk, D, device="cuda") * (0.35 if D == 64 else 1.0)).to(torch.bfloat16).transpose(1, 2)
        v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
        idx = torch.arange(S, device="cuda")
        keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
        mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
        st = ops.mask_structure(mask, S)
        rep = H // hk
        ke = k[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else k
        ve = v[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else v
        with torch.no_grad():
            ro, rl = torch.ops.aten._scaled_dot_product_efficient_attention(q, ke, ve, mask.expand(B, H, S, S), True, 0.0, False, scale=scale)[:2]
            rec = {"H": H, "D": D, "S": S}
            for kb in (16, 32, 64):
                got = ops.attn_fwd_exact(q, k, v, st, scale, key_block=kb)
                rec[f"kb{kb}"] = None if got is None else [nd(got[0], ro.transpose(1, 2)), nd(got[1], rl[..., :S])]
        print(json.dumps(rec), flush=True)
        res.append(rec)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_exact_keyblock_probe.json"), "w"), indent=1)
