import torch, torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters
B,H,KV,S,D=8,32,8,2048,128
q=torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16,requires_grad=True)
k=torch.randn(B,KV,S,D,device='cuda',dtype=torch.bfloat16,requires_grad=True)
v=torch.randn(B,KV,S,D,device='cuda',dtype=torch.bfloat16,requires_grad=True)
do=torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16)
def gqa():
    o=F.scaled_dot_product_attention(q,k,v,is_causal=True,enable_gqa=True); o.backward(do)
def rep():
    kk=k[:,:,None].expand(B,KV,H//KV,S,D).reshape(B,H,S,D); vv=v[:,:,None].expand(B,KV,H//KV,S,D).reshape(B,H,S,D)
    o=F.scaled_dot_product_attention(q,kk,vv,is_causal=True); o.backward(do)
for name,fn in (('enable_gqa',gqa),('repeat_kv',rep)):
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
        try:
            with sdpa_kernel(be): print(name, be, 'ms', round(bench(fn),3))
        except Exception as ex: print(name, be, 'failed', repr(ex)[:120])
    with sdpa_kernel([SDPBackend.EFFICIENT_ATTENTION, SDPBackend.FLASH_ATTENTION, SDPBackend.MATH], set_priority=True):
        print(name, 'priority efficient-first ms', round(bench(fn),3))
