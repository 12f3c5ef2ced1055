"""Which AOTriton kernel images does torch's SDPA launch for the two attention problems of the bit-identical paths?
Run under `rocprofv3 --kernel-trace --output-format csv`: the trace's grid / workgroup / LDS / VGPR columns identify the tuned
configuration (BLOCK_M, BLOCK_N, warps) among the images in torch/lib/aotriton.images/amd-gfx950/flash/*.aks2.
Also saves one small input / output set per problem (gpurun_out/r06/aot_case_<D>.pt) for offline work."""
import os, sys, torch, torch.nn.functional as F
torch.manual_seed(0)
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", os.environ.get("AR_TAG", "r06"))
os.makedirs(out, exist_ok=True)
for (B, H, S, D, hk) in ((8, 32, 2048, 128, 8), (8, 12, 2048, 64, 12)):
    q = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    if hk != H:     # repeat_kv: contiguous [B, H, S, D]
        k = torch.randn(B, hk, S, D, device="cuda", dtype=torch.bfloat16)[:, :, None].expand(B, hk, H // hk, S, D).reshape(B, H, S, D)
        v = torch.randn(B, hk, S, D, device="cuda", dtype=torch.bfloat16)[:, :, None].expand(B, hk, H // hk, S, D).reshape(B, H, S, D)
    else:
        k = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
        v = torch.randn(B, S, H, D, device="cuda", dtype=torch.bfloat16).transpose(1, 2)
    keep = torch.tril(torch.ones(S, S, device="cuda", dtype=torch.bool))[None, None].expand(B, 1, S, S).clone()
    keep[1, :, :, S - 100:] = False
    mask = keep.to(torch.bfloat16)            # the calibration flow's 0 / 1 additive bias
    scale = 1.0 if D == 64 else D ** -0.5
    with torch.no_grad():
        o_inf = F.scaled_dot_product_attention(q, k, v, attn_mask=mask, dropout_p=0.0, scale=scale, is_causal=False)
    ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
    o = F.scaled_dot_product_attention(ql, kl, vl, attn_mask=mask, dropout_p=0.0, scale=scale, is_causal=False)
    do = torch.randn_like(o)
    gq, gk, gv = torch.autograd.grad(o, (ql, kl, vl), do)
    torch.cuda.synchronize()
    print(D, "out", tuple(o.shape), o.stride(), "inference == training:", bool(torch.equal(o_inf, o)),
          "gq", gq.stride(), "gk", gk.stride(), "gv", gv.stride(), flush=True)
print("ok")
