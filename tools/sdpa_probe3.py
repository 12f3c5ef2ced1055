"""Which flash-attention libraries does this torch build expose on gfx950, and how fast are they for the Llama-3-8B
tuning shape (8 x 32 x 2048 x 128 causal, forward and forward+backward)?"""
import torch, torch.nn.functional as F
from torch.nn.attention import sdpa_kernel, SDPBackend


def bench(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / iters


print("torch", torch.__version__, "hip", torch.version.hip)
try:
    print("preferred_rocm_fa_library:", torch.backends.cuda.preferred_rocm_fa_library())
except Exception as ex:
    print("preferred_rocm_fa_library failed", repr(ex)[:200])
try:
    import aiter
    print("aiter importable", getattr(aiter, "__version__", "?"))
except Exception as ex:
    print("aiter not importable:", repr(ex)[:100])
print("torch.ops.aiter:", [n for n in dir(torch.ops.aiter)][:20] if hasattr(torch.ops, "aiter") else None)
B, H, S, D = 8, 32, 2048, 128
q = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
k = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
v = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16, requires_grad=True)
do = torch.randn(B, H, S, D, device="cuda", dtype=torch.bfloat16)
fl_f = 4 * B * H * S * S * D / 2 / 1e9


def fwd():
    with torch.no_grad():
        F.scaled_dot_product_attention(q, k, v, is_causal=True)


def fb():
    o = F.scaled_dot_product_attention(q, k, v, is_causal=True); o.backward(do)


for lib in ("aotriton", "ck", "default"):
    try:
        torch.backends.cuda.preferred_rocm_fa_library(lib)
        print("lib ->", lib, "now", torch.backends.cuda.preferred_rocm_fa_library())
    except Exception as ex:
        print("lib", lib, "not selectable:", repr(ex)[:160]); continue
    for be in (SDPBackend.FLASH_ATTENTION, SDPBackend.EFFICIENT_ATTENTION):
        try:
            with sdpa_kernel(be):
                tf, tb = bench(fwd), bench(fb)
            print(f"  {lib:8s} {str(be):40s} fwd {tf:.3f} ms ({fl_f / tf:.0f} TF/s)  fwd+bwd {tb:.3f} ms ({3.5 * fl_f / tb:.0f} TF/s)")
        except Exception as ex:
            print("  ", lib, be, "failed", repr(ex)[:160])
