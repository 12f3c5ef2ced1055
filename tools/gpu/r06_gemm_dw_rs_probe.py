"""k_gemm_dw6 with the operand pieces staged through registers (ar_gemm_dw_config(33)) against the LDS-DMA form (32) and hipBLASLt:
bit equality on small asymmetric problems (strided, accumulate, ragged K, split plans) and at Llama-3-8B's shapes, kernel times."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops, streamk
from auto_round_amd._lib import load
V = {"dma": 32, "rs": 33}
lib = load()
def use(v): lib.ar_gemm_dw_config(V[v], -1)
def ndiff(a, b): return int((a.contiguous().view(torch.int16) != b.contiguous().view(torch.int16)).sum())
def rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)
def timed(fn, reps):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn(); s.record()
    for _ in range(reps): fn()
    e.record(); e.synchronize()
    return s.elapsed_time(e) / reps
res = {"small": [], "shapes": []}
cases = [(256, 512, 256, False, False), (512, 1024, 2048, True, True), (1000, 512, 512, False, False), (96 + 37, 256, 256, False, False),
         (4096, 768, 768, False, False), (2048, 3072, 768, False, False), (16384 + 64, 512, 256, False, False)]
for K, M, N, strided, accum in cases:
    if strided:
        by, bx = rnd((K, M + 512), 1), rnd((K, N + 256), 2, 0.05)
        dY, X = by[:, 256:256 + M], bx[:, 256:256 + N]
    else:
        dY, X = rnd((K, M), 1), rnd((K, N), 2, 0.05)
    outs = {}
    for v in V:
        use(v)
        for split in ([True, False, 2] if K >= 1024 else [True, False]):
            old = rnd((M, N), 3, 0.5)
            out = old.clone()
            ok = ops.gemm_dw(dY, X, out, accumulate=accum, split=split)
            outs[(v, str(split))] = out if ok else None
    for split in ("True", "False", "2"):
        a, b = outs.get(("dma", split)), outs.get(("rs", split))
        if a is not None and b is not None:
            rec = dict(K=K, M=M, N=N, strided=strided, accumulate=accum, split=split, rs_vs_dma_differing=ndiff(a, b), numel=a.numel())
            res["small"].append(rec); print(json.dumps(rec), flush=True)
shapes = {"q_o": (4096, 4096), "down": (4096, 14336), "gate_up": (14336, 4096), "qkv_merged": (6144, 4096)}
K = 16384
for name, (M, N) in shapes.items():
    dY, X = rnd((K, M), 11, 0.02), rnd((K, N), 12)
    flops = 2.0 * M * N * K
    lib_out = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    torch.mm(dY.t(), X, out=lib_out)
    st = streamk.find_on_device(dY, X) if M * N >= 4096 * 14336 else None
    kcut = None if st is None else st[1]
    one, sk, fns = {}, {}, {"hipblaslt": lambda: torch.mm(dY.t(), X, out=lib_out)}
    for v in V:
        use(v)
        o = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        assert ops.gemm_dw(dY, X, o, split=False)
        one[v] = o
        for _ in range(3):
            o2 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            ops.gemm_dw(dY, X, o2, split=False)
            assert ndiff(o, o2) == 0, "two launches, different bits"
        if kcut is not None:
            o3 = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
            assert ops.gemm_dw_sk(dY, X, o3, kcut)
            sk[v] = o3
        def f1(v=v, o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda")):
            use(v); ops.gemm_dw(dY, X, o, split=False)
        fns[f"{v}_one_pass"] = f1
        if kcut is not None:
            def f3(v=v, o=torch.empty((M, N), dtype=torch.bfloat16, device="cuda")):
                use(v); ops.gemm_dw_sk(dY, X, o, kcut)
            fns[f"{v}_streamk"] = f3
    times = {k: [] for k in fns}
    for _ in range(3):
        for k, f in fns.items(): times[k].append(timed(f, 10))
    rec = dict(shape=name, M=M, N=N, K=K, one_pass_rs_vs_dma_differing=ndiff(one["rs"], one["dma"]), one_pass_rs_vs_library_differing=ndiff(one["rs"], lib_out))
    if kcut is not None:
        rec.update(streamk_rs_vs_dma_differing=ndiff(sk["rs"], sk["dma"]), streamk_rs_vs_library_differing=ndiff(sk["rs"], lib_out))
    for k, ts in times.items():
        ms = sorted(ts)[len(ts) // 2]
        rec[f"{k}_ms"] = round(ms, 4); rec[f"{k}_pflops"] = round(flops / ms / 1e12, 4)
    res["shapes"].append(rec); print(json.dumps(rec), flush=True)
    del dY, X, lib_out, one, sk, fns
    torch.cuda.empty_cache()
use("dma")
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "gemm_dw_rs_probe.json"), "w"), indent=1)
