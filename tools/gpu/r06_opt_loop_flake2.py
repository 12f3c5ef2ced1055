"""Round 6 probe: WHERE does a 200-iteration OPT-125M tune part from itself when its targets are identical (2 of 12 runs in
r06_opt_loop_flake.py)?  The real loop (`SignRoundQuantizer.quantize_block`, module path) run for I iterations on ONE minibatch with
learning rate 0: every iteration computes exactly the same thing, so every per-iteration checksum must repeat -- of each stage's forward
output, of the gradient arriving at each stage's output (tensor hooks), of the block's weight-gradient arena after the backward, of the
loss.  An iteration whose checksum differs names the op.

    python tools/gpu/r06_opt_loop_flake2.py [I]   ->  gpurun_out/r06/opt_loop_flake2.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd import ops  # noqa: E402
from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer  # noqa: E402
from auto_round_amd.schemes import apply_scheme, resolve_scheme  # noqa: E402
from auto_round_amd.testing import t3_fixture as fx  # noqa: E402


def main():
    I = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
    dev = torch.device("cuda:0")
    out = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "iterations": I}
    torch.use_deterministic_algorithms(True, warn_only=True)
    for fuse in (True, False):
        model = fx.build_model("opt125m").to(dev)
        for p in model.parameters():
            p.requires_grad_(False)
        tokens = fx.calib_tokens("opt125m", 16, 2048)
        block = fx.decoder_blocks(model)[0]
        apply_scheme(block, resolve_scheme("W4A16"))
        x0, others = fx.capture_block_inputs(model, block, tokens, dev)
        q = SignRoundQuantizer(SignRoundConfig(iters=I, batch_size=8, bits=4, sdpa_backend="auto", lr=0.0, minmax_lr=0.0, fuse_next_forward=fuse,
                                               not_use_best_mse=False), device=dev)
        y = q.calibrate_block(block, x0, others)
        stages = list(fx.OPT_STAGES[:-1])
        cols = ["fwd:" + s for s in stages] + ["grad:" + s for s in stages] + ["dW"]
        table = torch.zeros(I + 8, len(cols), dtype=torch.int64, device=dev)
        it = {"i": 0}

        def fwd_hook(name):
            ci, gi = cols.index("fwd:" + name), cols.index("grad:" + name)

            def f(mod, inp, o):
                t = o[0] if isinstance(o, tuple) else o
                table[it["i"], ci].copy_(fx.bits_checksum(t))
                if t.requires_grad:
                    def g(grad, row=it["i"]):
                        table[row, gi].copy_(fx.bits_checksum(grad))
                    t.register_hook(g)
            return f

        def core_pre(mod, args):
            t = args[0]
            table[it["i"], cols.index("fwd:attn_core")].copy_(fx.bits_checksum(t))
            if t.requires_grad:
                def g(grad, row=it["i"]):
                    table[row, cols.index("grad:attn_core")].copy_(fx.bits_checksum(grad))
                t.register_hook(g)

        real_update = ops.best_loss_update

        def update(total_loss, state, istate, i, loss_hist=None, **kw):
            for a in getattr(block, "_ar_arenas", []):
                table[it["i"], cols.index("dW")].copy_(fx.bits_checksum(a.dWq))
            it["i"] += 1
            return real_update(total_loss, state, istate, i, loss_hist=loss_hist, **kw)

        import auto_round_amd.quantizer as qmod

        hs = []
        # the hooks must sit on the modules that exist DURING tuning (the wrappers): install them from inside wrapper_block
        real_wrap = q.wrapper_block

        def wrap_and_hook(blk, *a, **kw):
            res = real_wrap(blk, *a, **kw)
            mods = dict(blk.named_modules())
            for s in stages:
                if s in mods:
                    hs.append(mods[s].register_forward_hook(fwd_hook(s)))
            hs.append(mods["self_attn.out_proj"].register_forward_pre_hook(core_pre))
            return res

        q.wrapper_block = wrap_and_hook
        ops.best_loss_update = update
        qmod.ops.best_loss_update = update
        try:
            q.quantize_block(block, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=None, index_schedule=[list(range(8))] * I)
        finally:
            ops.best_loss_update = real_update
            for h in hs:
                h.remove()
        torch.cuda.synchronize()
        t = table[:I].cpu().numpy()
        trace = np.asarray(q.last_stats["loss_trace"], dtype=np.float64)
        rec = {"iterations": int(it["i"]), "distinct_losses": int(len(np.unique(trace))), "loss_off_majority": int(len(trace) - np.unique(trace, return_counts=True)[1].max())}
        per = {}
        bad_rows = set()
        for ci, c in enumerate(cols):
            vals, cnt = np.unique(t[:, ci], return_counts=True)
            maj = vals[cnt.argmax()]
            rows = np.nonzero(t[:, ci] != maj)[0]
            per[c] = {"off_majority": int(len(rows)), "rows": [int(r) for r in rows[:10]]}
            bad_rows.update(int(r) for r in rows)
        rec["per_checksum"] = per
        # for every iteration with anything off: the first column (forward order, then backward order) that differs
        order = ["fwd:" + s for s in stages] + ["grad:" + s for s in reversed(stages)] + ["dW"]
        firsts = {}
        for r in sorted(bad_rows)[:20]:
            for c in order:
                ci = cols.index(c)
                vals, cnt = np.unique(t[:, ci], return_counts=True)
                if t[r, ci] != vals[cnt.argmax()]:
                    firsts[r] = c
                    break
        rec["first_differing_checksum_per_bad_iteration"] = firsts
        out["fuse_next_forward_%s" % fuse] = rec
        print("fuse_next_forward", fuse, {k: v for k, v in rec.items() if k != "per_checksum"}, flush=True)
        print("   ", {c: v["off_majority"] for c, v in per.items()}, flush=True)
        del model, block, x0, y, q
        torch.cuda.empty_cache()
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06", "opt_loop_flake2.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


if __name__ == "__main__":
    main()
