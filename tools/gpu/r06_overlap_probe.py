"""Does an HBM-bound elementwise kernel run in the shadow of an MFMA-bound GEMM when the two are issued on different HIP streams?
(What a side stream for the weight-gradient GEMMs of the backward pass could buy.)  k_gemm_dw6 (14336 x 4096 x 16384) next to
k_x_swiglu_bwd (16384 x 14336) and next to the library's dX GEMM: serial time, concurrent time."""
import json, os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda")
bf = torch.bfloat16
T, M, N = 16384, 14336, 4096
dY = (0.01 * torch.randn(T, M, device=dev)).to(bf)
X = torch.randn(T, N, device=dev).to(bf)
dW = torch.empty(M, N, dtype=bf, device=dev)
g = torch.randn(T, M, device=dev).to(bf); u = torch.randn(T, M, device=dev).to(bf); da = torch.randn(T, M, device=dev).to(bf)
W = (0.02 * torch.randn(M, N, device=dev)).to(bf)
side = torch.cuda.Stream()
def gemm(): ops.gemm_dw(dY, X, dW, split=False)
def elem(): ops.swiglu_bwd_exact(da, g, u, contract=True)
def dx(): torch.mm(dY, W)
def timed(f, n=10):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def both(a, b):
    def f():
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            a()
        b()
        torch.cuda.current_stream().wait_stream(side)
    return f
rec = {"dw_gemm_ms": timed(gemm), "swiglu_bwd_ms": timed(elem), "dx_gemm_ms": timed(dx)}
rec["dw_gemm_and_swiglu_bwd_concurrent_ms"] = timed(both(gemm, elem))
rec["dw_gemm_and_dx_gemm_concurrent_ms"] = timed(both(gemm, dx))
rec["dw_gemm_and_3x_swiglu_concurrent_ms"] = timed(both(gemm, lambda: (elem(), elem(), elem())))
print(json.dumps(rec))
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
json.dump(rec, open(os.path.join(out, "overlap_probe.json"), "w"), indent=1)
