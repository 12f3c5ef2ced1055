"""Round 6 probe: two full-recipe OPT-125M tuning runs (200 iterations, real learning rate, same seed) in one process, every library /
first-party call of every iteration checksummed on the device in call order; the first (iteration, call) at which run B differs from
run A is where the trajectories part.  P pairs per path -> a histogram of the op that parted them.

    python tools/gpu/r06_opt_pair_trace.py [P] [exact|module]   ->  gpurun_out/r06/opt_pair_trace_<path>.json
"""
import json
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F
import transformers

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd import ops  # noqa: E402
from auto_round_amd.autoround import loss_mask_ids  # noqa: E402
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer  # noqa: E402
from auto_round_amd.schemes import apply_scheme, resolve_scheme  # noqa: E402
from auto_round_amd.testing import t3_fixture as fx  # noqa: E402

MAXC, ITERS = 48, 200


def main():
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    path = sys.argv[2] if len(sys.argv) > 2 else "exact"
    dev = torch.device("cuda:0")
    torch.use_deterministic_algorithms(True, warn_only=True)
    tokens = fx.calib_tokens("opt125m", 128, 2048)
    ids = loss_mask_ids(tokens, None)
    q = SignRoundQuantizer(SignRoundConfig(iters=ITERS, batch_size=8, bits=4, sdpa_backend="auto", exact_rounding=(path == "exact")), device=dev)
    st = {"row": 0, "col": 0, "on": False, "table": None, "names": {}}

    def tap(name, t):
        if st["on"] and isinstance(t, torch.Tensor) and t.element_size() == 2 and st["col"] < MAXC and st["row"] < ITERS:
            st["names"].setdefault(st["col"], name)
            st["table"][st["row"], st["col"]].copy_(fx.bits_checksum(t))
            st["col"] += 1
        return t

    real = dict(linear=F.linear, sdpa=F.scaled_dot_product_attention, mm=torch.mm, grad=torch.autograd.grad, lnf=ops.layernorm_fwd_exact,
                lnb=ops.layernorm_bwd_exact, upd=ops.best_loss_update, gather=ops.gather_rows, thr=torch.ops.aten.threshold_backward)

    def linear(*a, **k):
        return tap("F.linear", real["linear"](*a, **k))

    def sdpa(*a, **k):
        return tap("sdpa_forward", real["sdpa"](*a, **k))

    def mm(*a, **k):
        o = real["mm"](*a, **k)
        tap("torch.mm", k.get("out", o))
        return o

    def grad(*a, **k):
        g = real["grad"](*a, **k)
        for j, t in enumerate(g):
            tap(f"autograd.grad[{j}] (attention / norm backward)", t)
        return g

    def lnf(*a, **k):
        r = real["lnf"](*a, **k)
        if r is not None:
            tap("layernorm_fwd_exact", r[0])
        return r

    def lnb(*a, **k):
        return tap("layernorm_bwd_exact", real["lnb"](*a, **k))

    def gather(*a, **k):
        st["on"] = True                      # the tuning loop has started (the plan proof never gathers)
        return real["gather"](*a, **k)

    def upd(total_loss, state, istate, i, loss_hist=None, **kw):
        for a in getattr(st["block"], "_ar_arenas", []):
            tap("dW arena (all six weight gradients)", a.dWq)
        st["row"] += 1
        st["col"] = 0
        return real["upd"](total_loss, state, istate, i, loss_hist=loss_hist, **kw)

    def one_run():
        model = fx.build_model("opt125m").to(dev)
        for p in model.parameters():
            p.requires_grad_(False)
        block = fx.decoder_blocks(model)[0]
        apply_scheme(block, resolve_scheme("W4A16"))
        x0, others = fx.capture_block_inputs(model, block, tokens, dev)
        y = q.calibrate_block(block, x0, others)
        st.update(row=0, col=0, on=False, table=torch.zeros(ITERS, MAXC, dtype=torch.int64, device=dev), block=block)
        transformers.set_seed(42)
        F.linear, F.scaled_dot_product_attention, torch.mm, torch.autograd.grad = linear, sdpa, mm, grad
        ops.layernorm_fwd_exact, ops.layernorm_bwd_exact, ops.best_loss_update, ops.gather_rows = lnf, lnb, upd, gather
        try:
            q.quantize_block(block, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=ids)
        finally:
            F.linear, F.scaled_dot_product_attention, torch.mm, torch.autograd.grad = real["linear"], real["sdpa"], real["mm"], real["grad"]
            ops.layernorm_fwd_exact, ops.layernorm_bwd_exact, ops.best_loss_update, ops.gather_rows = real["lnf"], real["lnb"], real["upd"], real["gather"]
        torch.cuda.synchronize()
        return st["table"].cpu().numpy(), fx.sha(y), np.asarray(q.last_stats["loss_trace"])

    ref, ysha, tr0 = one_run()            # (includes the plan proof)
    ref, ysha, tr0 = one_run()
    hist, parted, details = {}, 0, []
    for p in range(P):
        t, ys, tr = one_run()
        if ys != ysha:
            details.append({"pair": p, "targets_differ": True})
            continue
        diff = np.argwhere(t != ref)
        if diff.size == 0:
            continue
        parted += 1
        it, c = map(int, diff[0])                       # argwhere is row-major: the first differing (iteration, call)
        name = f"{c}:{st['names'].get(c, '?')}"
        hist[name] = hist.get(name, 0) + 1
        details.append({"pair": p, "iteration": it, "call": name})
    rec = {"path": path, "exact_block": bool(q.last_exact), "pairs": P, "parted": parted, "first_differing_call": hist, "details": details,
           "call_order": [st["names"].get(c, "?") for c in range(max(st["names"]) + 1)]}
    print(json.dumps(rec), flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06", f"opt_pair_trace_{path}.json"), "w") as f:
        json.dump(rec, f, indent=1, default=str)


if __name__ == "__main__":
    main()
