"""A few launches of the exact attention kernels (forward, dK / dV, dQ) and of the library kernels they replace, at the two minibatch
shapes, for one `rocprofv3 --pmc` pass (MFMA busy, wait breakdown, LDS bank conflicts, VALU activity):

    bash tools/gpu/run.sh pmc attn "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" -- tools/gpu/r06_pmc_attn_kernels.py
    python tools/gpu/r06_pmc_attn_kernels.py --summarise gpurun_out/<tag>/attn_counter_collection.csv > profiles/r06_pmc_attn_kernels.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        import r04_pmc_kernels as r4

        print(json.dumps(r4.summarise(sys.argv[2], names=("k_xattn", "attn_fwd", "bwd_kernel", "bwd_preprocess")), indent=1))
        return
    import torch
    import torch.nn.functional as F

    from auto_round_amd import ops

    torch.manual_seed(0)
    for (B, H, S, D, hk, scale, std) in ((8, 32, 2048, 128, 8, 128 ** -0.5, 1.0), (8, 12, 2048, 64, 12, 1.0, 0.35)):
        q = (torch.randn(B, S, H, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
        k = (torch.randn(B, S, hk, D, device="cuda") * std).to(torch.bfloat16).transpose(1, 2)
        v = torch.randn(B, S, hk, D, device="cuda").to(torch.bfloat16).transpose(1, 2)
        idx = torch.arange(S, device="cuda")
        keep = (idx[None, :] <= idx[:, None]) & (idx[None, :] < S - 1)
        mask = keep.to(torch.bfloat16)[None, None].expand(B, 1, S, S).contiguous()
        st = ops.mask_structure(mask, S)
        da = (torch.randn(B, S, H, D, device="cuda") * 0.02).to(torch.bfloat16)
        rep = H // hk
        for _ in range(2):
            with torch.no_grad():
                o, lse = ops.attn_fwd_exact(q, k, v, st, scale)
                ops.attn_bwd_exact(q, k, v, o, lse, da, st, scale)
            ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
            ke = kl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else kl
            ve = vl[:, :, None].expand(B, hk, rep, S, D).reshape(B, H, S, D) if rep > 1 else vl
            ao = F.scaled_dot_product_attention(ql, ke, ve, attn_mask=mask, dropout_p=0.0, is_causal=False, scale=scale).transpose(1, 2).contiguous()
            torch.autograd.grad(ao, (ql, kl, vl), da)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
