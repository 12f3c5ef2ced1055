import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
torch.manual_seed(0)
from transformers.models.qwen3.modeling_qwen3 import Qwen3RMSNorm
def nd(a, b):
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return int((a.contiguous().view(it) != b.contiguous().view(it)).sum())
for (T, H, D) in ((16384, 32, 128), (16384, 8, 128), (4096, 16, 128)):
    norm = Qwen3RMSNorm(D, eps=1e-6).to("cuda").to(torch.bfloat16)
    with torch.no_grad():
        norm.weight.copy_((1 + 0.1 * torch.randn(D, device="cuda")).to(torch.bfloat16))
    x = (torch.randn(T, H * D, device="cuda") * 1.3).to(torch.bfloat16)
    xl = x.detach().requires_grad_(True)
    y = norm(xl.view(1, T, H, D))
    g = (torch.randn(1, T, H, D, device="cuda") * 0.1).to(torch.bfloat16)
    (gx,) = torch.autograd.grad(y, xl, g)
    rows = x.view(T * H, D)
    want_r = torch.rsqrt(rows.float().pow(2).mean(-1, keepdim=True) + 1e-6).view(-1)
    res = ops.rmsnorm_fwd_exact(rows, norm.weight.detach(), 1e-6)
    if res is None:
        print(T, H, D, "refused"); continue
    yy, rstd, _ = res
    dx = ops.rmsnorm_bwd_exact(g.view(T * H, D).contiguous(), rows, norm.weight.detach(), rstd)
    print(T, H, D, "y", nd(yy.view(1, T, H, D), y.detach()), "rstd", nd(rstd, want_r), "dx", nd(dx.view(T, H * D), gx))
