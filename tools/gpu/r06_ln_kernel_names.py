import torch
torch.manual_seed(0)
for M in (16384, 32768, 65536):
    x = torch.randn(M, 768, device="cuda").to(torch.bfloat16).float().requires_grad_(True)
    w = torch.randn(768, device="cuda").to(torch.bfloat16).float(); b = torch.randn(768, device="cuda").to(torch.bfloat16).float()
    y = torch.nn.functional.layer_norm(x, (768,), w, b, 1e-5)
    g, = torch.autograd.grad(y, x, torch.randn_like(y))
    torch.cuda.synchronize()
print("ok")
