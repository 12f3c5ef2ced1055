"""ar_attn_bwd_exact against autograd of torch's attention (aten::_scaled_dot_product_efficient_attention_backward = AOTriton
bwd_preprocess + bwd_kernel_dk_dv + bwd_kernel_dq with an additive bias) on the two problems of the bit-identical paths."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
torch.manual_seed(0)
res = []
shapes = ((8, 32, 2048, 128, 8, 128 ** -0.5, 1.0), (8, 12, 2048, 64, 12, 1.0, 0.35))
for (B, H, S, D, hk, scale, std) in shapes:
    q = (torch.randn(B, S, H, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
    k = (torch.randn(B, S, hk, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
    rep = H // hk
    valid = S - 1
    idx = torch.arange(S, device="cuda")
    keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < valid)
    mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
    st = ops.mask_structure(mask, S)
    ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
    ke = kl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else kl
    ve = vl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else vl
    if rep > 1:
        ke.retain_grad(); ve.retain_grad()
    o = torch.nn.functional.scaled_dot_product_attention(ql, ke, ve, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale)
    ao = o.transpose(1, 2).contiguous()                     # [B, S, H, D], as transformers hands it on
    da = (torch.randn(B, S, H * D, device="cuda") * 0.02).to(torch.bfloat16)
    gq, gk, gv = torch.autograd.grad(ao, (ql, kl, vl), da.view(B, S, H, D), retain_graph=True)
    gke = gve = None
    if rep > 1:
        gke, gve = torch.autograd.grad(ao, (ke, ve), da.view(B, S, H, D), retain_graph=True)
    torch.cuda.synchronize()
    with torch.no_grad():
        mo, mlse = ops.attn_fwd_exact(q, k, v, st, scale)
        got = ops.attn_bwd_exact(q, k, v, mo, mlse, da.view(B, S, H, D), st, scale)
    assert got is not None
    dq, dke, dve = got                                      # [B, S, H, D]
    torch.cuda.synchronize()
    def ndiff(a, b):
        a, b = a.contiguous(), b.contiguous()
        return int((a.view(torch.int16) != b.view(torch.int16)).sum())
    rec = {"shape": [B, H, S, D, hk], "fwd_out_differ": ndiff(mo, ao.detach()), "values": dq.numel(),
           "dq_differ": ndiff(dq, gq.transpose(1, 2)), "gq_stride": list(gq.stride()), "gk_stride": list(gk.stride())}
    if rep > 1:
        rec["dk_expanded_differ"] = ndiff(dke, gke.transpose(1, 2))
        rec["dv_expanded_differ"] = ndiff(dve, gve.transpose(1, 2))
        # the group sum as autograd's expand backward does it
        dks = dke.view(B, S, hk, rep, D).transpose(1, 2).transpose(2, 3)      # [B, hk, rep, S, D] view
        dk_sum = dke.transpose(1, 2).reshape(B, hk, rep, S, D).sum(2)
        dv_sum = dve.transpose(1, 2).reshape(B, hk, rep, S, D).sum(2)
        rec["dk_differ"] = ndiff(dk_sum, gk)
        rec["dv_differ"] = ndiff(dv_sum, gv)
        dk_sum2 = dke.view(B, S, hk, rep, D).sum(3)
        rec["dk_differ_token_major_sum"] = ndiff(dk_sum2.transpose(1, 2), gk)
    else:
        rec["dk_differ"] = ndiff(dke, gk.transpose(1, 2))
        rec["dv_differ"] = ndiff(dve, gv.transpose(1, 2))
    rec["max_abs_dq"] = float((dq.float() - gq.transpose(1, 2).float()).abs().max())
    for name, mine, ref in (("dq", dq, gq.transpose(1, 2)), ("dk", dke, (gke if rep > 1 else gk).transpose(1, 2)), ("dv", dve, (gve if rep > 1 else gv).transpose(1, 2))):
        bad = (mine.contiguous().view(torch.int16) != ref.contiguous().view(torch.int16))
        if bool(bad.any()):
            nz = bad.nonzero()
            rec[name + "_bad_first"] = nz[:6].tolist()
            rec[name + "_bad_tokens_mod32"] = sorted(set((nz[:2000, 1] % 32).tolist()))[:40]
            rec[name + "_bad_rows"] = int(bad.any(-1).sum())
    print(json.dumps(rec), flush=True)
    def tm(f, n=5):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    rec["ms_library_bwd"] = tm(lambda: torch.autograd.grad(ao, (ql, kl, vl), da.view(B, S, H, D), retain_graph=True))
    with torch.no_grad():
        rec["ms_first_party_bwd"] = tm(lambda: ops.attn_bwd_exact(q, k, v, mo, mlse, da.view(B, S, H, D), st, scale))
    print(json.dumps(rec), flush=True)
    res.append(rec)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_exact_bwd_probe.json"), "w"), indent=1)
