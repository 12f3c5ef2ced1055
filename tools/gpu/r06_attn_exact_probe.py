"""ar_attn_fwd_exact against torch's attention (aten::_scaled_dot_product_efficient_attention = AOTriton attn_fwd with an additive bias)
on the two problems of the bit-identical paths: differing output values / log-sum-exp values, and both kernels' times."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
torch.manual_seed(0)
res = []
for (B, H, S, D, hk, scale, std) in ((8, 32, 2048, 128, 8, 128 ** -0.5, 1.0), (8, 12, 2048, 64, 12, 1.0, 0.35), (2, 8, 512, 128, 2, 128 ** -0.5, 2.0),
                                       (3, 6, 384, 64, 6, 1.0, 0.5)):
    q = (torch.randn(B, S, H, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
    k = (torch.randn(B, S, hk, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
    v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
    rep = H // hk
    ke = k[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else k
    ve = v[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else v
    valid = S - 1
    idx = torch.arange(S, device="cuda")
    keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < valid)
    mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
    st = ops.mask_structure(mask, S)
    assert st is not None, "mask structure"
    with torch.no_grad():
        got = ops.attn_fwd_exact(q, k, v, st, scale)
        torch.cuda.synchronize(); print("first-party ran", flush=True)
        sdpa = torch.nn.functional.scaled_dot_product_attention(q, ke, ve, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale)
        torch.cuda.synchronize(); print("sdpa ran", flush=True)
        mask_e = mask.expand(B, H, S, S)
        ref_o, ref_lse = torch.ops.aten._scaled_dot_product_efficient_attention(q, ke, ve, mask_e, True, 0.0, False, scale=scale)[:2]
        torch.cuda.synchronize()
    assert got is not None
    o, lse = got
    ro = ref_o.transpose(1, 2)          # [B, S, H, D]
    d_o = int((o.view(torch.int16) != ro.contiguous().view(torch.int16)).sum())
    d_l = int((lse.view(torch.int32) != ref_lse[..., :S].contiguous().view(torch.int32)).sum())
    d_s = int((sdpa.transpose(1, 2).contiguous().view(torch.int16) != o.view(torch.int16)).sum())
    rec = {"shape": [B, H, S, D, hk], "out_values": o.numel(), "out_differ": d_o, "lse_values": lse.numel(), "lse_differ": d_l,
           "sdpa_vs_mine_differ": d_s, "ref_out_stride": list(ref_o.stride()), "ref_lse_shape": list(ref_lse.shape),
           "max_abs_out_diff": float((o.float() - ro.float()).abs().max()), "max_abs_lse_diff": float((lse - ref_lse[..., :S]).abs().max())}
    if d_l:
        bad = (lse.view(torch.int32) != ref_lse[..., :S].contiguous().view(torch.int32)).nonzero()[:5].tolist()
        rec["lse_first_bad"] = [(i, float(lse[tuple(i)]), float(ref_lse[tuple(i)])) for i in bad]
    if d_o:
        badm = (o.view(torch.int16) != ro.contiguous().view(torch.int16))
        rec["out_bad_rows"] = int(badm.any(-1).sum())
        rec["out_first_bad"] = badm.nonzero()[:5].tolist()
    # times
    def tm(f, n=10):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    with torch.no_grad():
        rec["ms_library"] = tm(lambda: torch.ops.aten._scaled_dot_product_efficient_attention(q, ke, ve, mask_e, True, 0.0, False, scale=scale))
        rec["ms_first_party"] = tm(lambda: ops.attn_fwd_exact(q, k, v, st, scale))
    print(json.dumps(rec), flush=True)
    res.append(rec)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06")
os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "attn_exact_probe.json"), "w"), indent=1)
