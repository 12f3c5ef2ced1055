"""Does PyTorch's TunableOp find faster hipBLASLt / rocBLAS solutions than the default heuristic for the forward / input-gradient
GEMMs of the Llama-3-8B block (the ones this repository leaves to the library)?  Times each call form used by
auto_round_amd/fused_block.py at the tuning minibatch (8 x 2048 tokens) with the default selection, then with tuning enabled, and
writes the tuned selections to a CSV.  Developer tool: python tools/tunableop_probe.py [--out results.csv]"""
import argparse
import json
import os
import sys

import torch
import torch.nn.functional as F


def timeit(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default="gpurun_out/tunableop_llama8b.csv")
    ap.add_argument("--tokens", type=int, default=16384)
    a = ap.parse_args()
    T, H, KV, Fd = a.tokens, 4096, 1024, 14336
    dev, dt = "cuda", torch.bfloat16
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s: torch.randn(*s, device=dev, generator=g).to(dt)
    Wqkv, Wo, Wgu, Wd = r(H + 2 * KV, H) * 0.02, r(H, H) * 0.02, r(2 * Fd, H) * 0.02, r(H, Fd) * 0.02
    xh, xf, res = r(T, H), r(T, Fd), r(T, H)
    dqkv, dgu = r(T, H + 2 * KV), r(T, 2 * Fd)
    calls = {
        "fwd_qkv  F.linear [T,4096]x[6144,4096]^T": (lambda: F.linear(xh, Wqkv), 2 * T * H * (H + 2 * KV)),
        "fwd_o    addmm(res, [T,4096], Wo^T)": (lambda: torch.addmm(res, xh, Wo.t()), 2 * T * H * H),
        "fwd_gu   F.linear [T,4096]x[28672,4096]^T": (lambda: F.linear(xh, Wgu), 2 * T * H * 2 * Fd),
        "fwd_down addmm(res, [T,14336], Wd^T)": (lambda: torch.addmm(res, xf, Wd.t()), 2 * T * H * Fd),
        "dx_down  mm([T,4096], Wd[4096,14336])": (lambda: torch.mm(xh, Wd), 2 * T * H * Fd),
        "dx_gu    mm([T,28672], Wgu[28672,4096])": (lambda: torch.mm(dgu, Wgu), 2 * T * H * 2 * Fd),
        "dx_o     mm([T,4096], Wo[4096,4096])": (lambda: torch.mm(xh, Wo), 2 * T * H * H),
    }
    base = {k: timeit(fn) for k, (fn, _) in calls.items()}
    import torch.cuda.tunable as tun
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    tun.set_filename(a.out)
    tun.enable(True)
    tun.tuning_enable(True)
    tun.set_max_tuning_duration(400)        # ms per solution
    tun.set_max_tuning_iterations(30)
    for k, (fn, _) in calls.items():
        fn()
        torch.cuda.synchronize()
    tun.tuning_enable(False)
    tuned = {k: timeit(fn) for k, (fn, _) in calls.items()}
    try:
        tun.write_file(a.out)
    except Exception:
        pass
    rows = []
    for k, (_, fl) in calls.items():
        rows.append({"call": k, "default_ms": round(base[k], 4), "tuned_ms": round(tuned[k], 4),
                     "default_pflops": round(fl / base[k] / 1e12, 3), "tuned_pflops": round(fl / tuned[k] / 1e12, 3)})
    print(json.dumps({"tokens": T, "sum_default_ms": round(sum(base.values()), 3), "sum_tuned_ms": round(sum(tuned.values()), 3), "rows": rows,
                      "results": [list(map(str, x)) for x in tun.get_results()]}, indent=1))


if __name__ == "__main__":
    sys.exit(main())
