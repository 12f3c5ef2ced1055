import torch
import torch.nn.functional as F
def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s=torch.cuda.Event(enable_timing=True); e=torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e)/iters
B,H,KV,S,D=8,32,8,2048,128
q=torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16,requires_grad=True)
k=torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16,requires_grad=True)
v=torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16,requires_grad=True)
do=torch.randn(B,H,S,D,device='cuda',dtype=torch.bfloat16)
def run():
    o=F.scaled_dot_product_attention(q,k,v,is_causal=True)
    o.backward(do)
print('default fwd+bwd ms', bench(run))
def fwd():
    with torch.no_grad(): F.scaled_dot_product_attention(q,k,v,is_causal=True)
print('default fwd ms', bench(fwd))
try:
    print('fa libs', torch.backends.cuda.preferred_rocm_fa_library())
    torch.backends.cuda.preferred_rocm_fa_library("ck")
    print('ck fwd+bwd ms', bench(run)); print('ck fwd ms', bench(fwd))
except Exception as ex:
    print('ck not available:', repr(ex)[:200])
from torch.nn.attention import sdpa_kernel, SDPBackend
for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
    try:
        with sdpa_kernel(be):
            print(be, 'fwd+bwd ms', bench(run))
    except Exception as ex:
        print(be, 'failed', repr(ex)[:150])
