"""Are the library operations of a Mixtral-8x7B expert pass (module path) reproducible call after call at irregular token counts?
F.linear forward, its input gradient and its weight gradient for the w1 / w3 (4096 -> 14336) and w2 (14336 -> 4096) projections at
per-expert row counts like a routed minibatch's, index_add_ / the indexing backward; each repeated and compared bit for bit."""
import json, os, sys, torch
torch.manual_seed(0)
torch.use_deterministic_algorithms(True, warn_only=True)
dev = "cuda"
res = []
def bits(t): return t.contiguous().view(torch.int16)
counts = [4100, 3900, 4500, 3700, 4096, 4200, 4000, 4272, 4101, 3999, 4333, 2047, 6001, 5123, 7777, 1]
H, F_ = 4096, 14336
W1 = (torch.randn(F_, H, device=dev) * 0.02).to(torch.bfloat16)
W2 = (torch.randn(H, F_, device=dev) * 0.02).to(torch.bfloat16)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 25
for M in counts:
    x = torch.randn(M, H, device=dev).to(torch.bfloat16)
    g1 = (torch.randn(M, F_, device=dev) * 0.01).to(torch.bfloat16)
    h = torch.randn(M, F_, device=dev).to(torch.bfloat16)
    g2 = (torch.randn(M, H, device=dev) * 0.01).to(torch.bfloat16)
    ops_ = {
        "w1 forward  x @ W1^T": lambda: torch.nn.functional.linear(x, W1),
        "w1 dX       g @ W1": lambda: g1 @ W1,
        "w1 dW       g^T @ x": lambda: g1.t() @ x,
        "w2 forward  h @ W2^T": lambda: torch.nn.functional.linear(h, W2),
        "w2 dX       g @ W2": lambda: g2 @ W2,
        "w2 dW       g^T @ h": lambda: g2.t() @ h,
    }
    for name, f in ops_.items():
        ref = f().clone()
        bad = 0
        for i in range(N):
            if i % 2:
                torch.empty(1 << 22, device=dev).normal_()          # other work in between
            bad += int(not torch.equal(bits(f()), bits(ref)))
        rec = {"rows": M, "op": name, "calls": N, "calls_differing": bad}
        if bad:
            print(json.dumps(rec), flush=True)
        res.append(rec)
    print("rows", M, "done", flush=True)
# index ops of the expert loop
T = 16384
hs = torch.randn(T, H, device=dev).to(torch.bfloat16).requires_grad_(True)
idx = torch.randperm(T, device=dev)[:4100]
cur = torch.randn(4100, H, device=dev).to(torch.bfloat16)
def index_add():
    out = torch.zeros(T, H, device=dev, dtype=torch.bfloat16)
    out.index_add_(0, idx, cur)
    out.index_add_(0, torch.flip(idx, [0])[:4000], cur[:4000])
    return out
def gather_bwd():
    y = hs[idx]
    (g,) = torch.autograd.grad(y, hs, cur)
    return g
for name, f in (("index_add_ (two experts)", index_add), ("backward of hidden[token_idx]", gather_bwd)):
    ref = f().clone(); bad = 0
    for i in range(N):
        bad += int(not torch.equal(bits(f()), bits(ref)))
    rec = {"op": name, "calls": N, "calls_differing": bad}
    print(json.dumps(rec), flush=True); res.append(rec)
print("TOTAL differing:", sum(r["calls_differing"] for r in res), "of", sum(r["calls"] for r in res), "calls")
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "moe_ops_repeat.json"), "w"), indent=1)
