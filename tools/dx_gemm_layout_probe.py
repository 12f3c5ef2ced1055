import torch, json, sys
dev = torch.device("cuda")
T = 16384
def tm(f, n=10):
    for _ in range(3): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n
out = []
for name, O, I in (("o", 4096, 4096), ("gate_up", 28672, 4096), ("down", 4096, 14336), ("qkv", 6144, 4096)):
    dY = torch.randn(T, O, device=dev, dtype=torch.bfloat16)
    W = torch.randn(O, I, device=dev, dtype=torch.bfloat16) * 0.02
    Wt = W.t().contiguous()
    dX = torch.empty(T, I, device=dev, dtype=torch.bfloat16)
    fl = 2.0 * T * O * I
    t_nn = tm(lambda: torch.mm(dY, W, out=dX))
    t_tn = tm(lambda: torch.mm(dY, Wt.t(), out=dX))
    t_tr = tm(lambda: Wt.copy_(W.t()))
    X = torch.randn(T, I, device=dev, dtype=torch.bfloat16)
    Y = torch.empty(T, O, device=dev, dtype=torch.bfloat16)
    t_fw = tm(lambda: torch.mm(X, W.t(), out=Y))
    r = dict(layer=name, out=O, inp=I, dx_nn_ms=t_nn, dx_tn_ms=t_tn, transpose_ms=t_tr, fwd_ms=t_fw,
             dx_nn_pf=fl / t_nn / 1e12, dx_tn_pf=fl / t_tn / 1e12, fwd_pf=fl / t_fw / 1e12)
    print(json.dumps(r)); out.append(r)
    del dY, W, Wt, dX, X, Y
json.dump(out, open(sys.argv[1] if len(sys.argv) > 1 else "dx_gemm_layout_probe.json", "w"), indent=1)
