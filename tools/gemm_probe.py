import torch, torch.nn.functional as F
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters
T=16384
for out_f,in_f in ((14336,4096),(4096,14336),(4096,4096),(1024,4096)):
    dy=torch.randn(T,out_f,device='cuda',dtype=torch.bfloat16); x=torch.randn(T,in_f,device='cuda',dtype=torch.bfloat16)
    w=torch.randn(out_f,in_f,device='cuda',dtype=torch.bfloat16); o=torch.empty(out_f,in_f,device='cuda',dtype=torch.bfloat16)
    fl=2*T*out_f*in_f/1e9
    t1=bench(lambda: torch.mm(dy.t(), x, out=o))
    dyT=dy.t().contiguous(); xT=x.t().contiguous()
    t2=bench(lambda: torch.mm(dyT, x, out=o))
    t3=bench(lambda: F.linear(dyT, xT))
    t4=bench(lambda: (dy.t().contiguous(), x.t().contiguous()))
    t5=bench(lambda: F.linear(x, w))          # forward
    t6=bench(lambda: torch.mm(dy, w))         # dX
    t7=bench(lambda: torch.mm(x.t(), dy))     # dW^T
    print(f"out={out_f} in={in_f}: dW mm(dy.t,x) {t1:.3f}ms {fl/t1:.0f}TF | mm(dyT_c,x) {t2:.3f} {fl/t2:.0f}TF | linear(dyT_c,xT_c) {t3:.3f} {fl/t3:.0f}TF | 2 transposes {t4:.3f}ms | fwd {t5:.3f} {fl/t5:.0f}TF | dX {t6:.3f} {fl/t6:.0f}TF | dW^T mm(x.t,dy) {t7:.3f} {fl/t7:.0f}TF")
