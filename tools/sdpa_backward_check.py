"""Which torch SDPA backends return correct gradients on this box?  For causal bf16 attention at several sequence lengths and both
operand layouts (token-major [B,S,H,D] viewed as [B,H,S,D] -- what transformers and this repository pass -- and contiguous
[B,H,S,D]) the gradients of each backend are compared with fp32 autograd of the exact softmax(QK^T)V.  Found with torch 2.10 +
ROCm 7.2 on MI355X: the "efficient" backend's backward is wrong (relative error ~1, NaNs) for token-major operands when
S % 256 == 128 and S > 128; auto_round_amd/attention.py routes those lengths to the flash kernels.
    python tools/sdpa_backward_check.py > profiles/r0N_sdpa_backward_check.json"""
import json
import math

import torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel


def main():
    B, H, D = 1, 8, 128
    rows = []
    for layout in ("token_major", "head_major"):
        for S in (128, 256, 384, 512, 640, 768, 896, 1024, 2048):
            torch.manual_seed(S)
            if layout == "token_major":
                q, k, v, do = (torch.randn(B, S, H, D, device="cuda").to(torch.bfloat16).transpose(1, 2) for _ in range(4))
            else:
                q, k, v, do = (torch.randn(B, H, S, D, device="cuda").to(torch.bfloat16) for _ in range(4))
            qf, kf, vf = (t.float().detach().requires_grad_(True) for t in (q, k, v))
            sc = (qf @ kf.transpose(-1, -2)) / math.sqrt(D)
            sc = sc.masked_fill(~torch.ones(S, S, device="cuda", dtype=torch.bool).tril(), float("-inf"))
            exact = torch.autograd.grad(torch.softmax(sc, -1) @ vf, (qf, kf, vf), do.float())
            for name, be in (("efficient", SDPBackend.EFFICIENT_ATTENTION), ("flash", SDPBackend.FLASH_ATTENTION), ("math", SDPBackend.MATH)):
                ql, kl, vl = (t.detach().requires_grad_(True) for t in (q, k, v))
                try:
                    with sdpa_kernel([be]):
                        o = F.scaled_dot_product_attention(ql, kl, vl, is_causal=True)
                    g = torch.autograd.grad(o, (ql, kl, vl), do)
                    err = [float(((a.float() - b).abs().max() / b.abs().max()).item()) for a, b in zip(g, exact)]
                    rows.append({"layout": layout, "S": S, "backend": name, "rel_err_dq_dk_dv": [round(e, 4) if e == e else None for e in err],
                                 "ok": all(e == e and e < 0.05 for e in err)})
                except Exception as e:  # backend not available for this problem
                    rows.append({"layout": layout, "S": S, "backend": name, "error": str(e)[:120]})
    print(json.dumps({"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "heads": H, "head_dim": D, "rows": rows}, indent=1))


if __name__ == "__main__":
    main()
