"""Round 6 probe (VERDICT r05 weak #1 / next #1b): which library op makes the OPT-125M block's fp forward (the TARGETS the
reference hands to quantize_block) differ between runs / boxes?

Runs the seeded OPT-125M-dimension block of tests/golden/t3s_opt125m_w4g128.npz over its 128 x 2048 calibration inputs R times in
the module path's own minibatches and keeps a device-side checksum of every stage (LayerNorm, q / k / v projections, attention core,
out_proj, fc1, fc2, block output) per minibatch -- no host synchronisation inside a pass (the failure was seen on asynchronous
calls), one read at the end.  Variants: synchronise after every op (`sync`), SDPA backend pinned, mask broadcast instead of
materialised, deterministic-algorithms mode off.  Then the full 200-iteration tune, T times.

    python tools/gpu/r06_opt_determinism.py [R] [T]   ->  gpurun_out/r06/opt_determinism.json
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer  # noqa: E402
from auto_round_amd.schemes import apply_scheme, resolve_scheme  # noqa: E402
from auto_round_amd.testing import t3_fixture as fx  # noqa: E402

STAGES = ["self_attn_layer_norm", "self_attn.q_proj", "self_attn.k_proj", "self_attn.v_proj", "attn_core", "self_attn.out_proj",
          "final_layer_norm", "fc1", "fc2", "block_out"]


def checksum(t: torch.Tensor) -> torch.Tensor:
    """order-sensitive 64-bit checksum of a tensor's bits, computed on the device, asynchronously"""
    b = t.detach().contiguous().view(torch.int16).reshape(-1).to(torch.int64) & 0xFFFF
    w = (torch.arange(b.numel(), device=b.device, dtype=torch.int64) % 65521) + 1
    return (b * w).sum()


def one_pass(q, block, x0, others, sync=False, bs=8):
    rec = {s: [] for s in STAGES}
    hs = []

    def out_hook(name):
        def f(mod, inp, out):
            o = out[0] if isinstance(out, tuple) else out
            rec[name].append(checksum(o))
            if sync:
                torch.cuda.synchronize()
        return f

    def pre_hook(mod, args):
        rec["attn_core"].append(checksum(args[0]))
        if sync:
            torch.cuda.synchronize()

    mods = dict(block.named_modules())
    for s in STAGES:
        if s in mods:
            hs.append(mods[s].register_forward_hook(out_hook(s)))
    hs.append(mods["self_attn.out_proj"].register_forward_pre_hook(pre_hook))
    outs = []
    try:
        with torch.no_grad():
            for b0 in range(0, x0.shape[0], bs):
                y = q.block_forward(block, x0[b0:b0 + bs], others)
                rec["block_out"].append(checksum(y))
                outs.append(y)
    finally:
        for h in hs:
            h.remove()
    return rec, torch.cat(outs, 0)


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 12
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    out = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "R": R}
    dev = torch.device("cuda:0")
    fixp = os.path.join(ROOT, "tests", "golden", "t3s_opt125m_w4g128.npz")
    m = json.loads(str(np.load(fixp, allow_pickle=False)["meta"]))
    out["fixture_y_sha"] = m["y_sha"]
    model = fx.build_model("opt125m").to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    tokens = fx.calib_tokens("opt125m", m["nsamples"], m["seqlen"])
    block = fx.decoder_blocks(model)[0]
    apply_scheme(block, resolve_scheme(m["scheme"]))
    x0, others = fx.capture_block_inputs(model, block, tokens, dev)
    out["inputs_identical"] = fx.sha(x0) == m["x_sha"]
    out["others"] = {k: (list(v.shape), str(v.dtype)) if isinstance(v, torch.Tensor) else str(type(v)) for k, v in others.items()}

    def variant(name, det=True, materialise=True, backend="auto", sync=False, reps=R):
        torch.use_deterministic_algorithms(det, warn_only=True)
        q = SignRoundQuantizer(SignRoundConfig(iters=0, batch_size=8, bits=4, sdpa_backend=backend, materialise_shared_rows=materialise), device=dev)
        sums, yshas = [], []
        for _ in range(reps):
            rec, y = one_pass(q, block, x0, others, sync=sync)
            sums.append(rec)
            yshas.append(y)
        torch.cuda.synchronize()
        ysha = [fx.sha(y) for y in yshas]
        del yshas
        table = {s: np.array([[int(c) for c in r[s]] for r in sums]) for s in STAGES}          # [reps, minibatches]
        per_stage = {}
        for s, a in table.items():
            # for every minibatch: how many passes differ from the majority value
            bad = 0
            where = []
            for j in range(a.shape[1]):
                vals, cnt = np.unique(a[:, j], return_counts=True)
                if len(vals) > 1:
                    bad += int(a.shape[0] - cnt.max())
                    where.append(j)
            per_stage[s] = dict(calls=int(a.size), calls_off_majority=bad, minibatches_affected=where[:16])
        first_bad = next((s for s in STAGES if per_stage[s]["calls_off_majority"]), None)
        res = dict(distinct_targets=len(set(ysha)), targets_match_fixture=sum(1 for h in ysha if h == m["y_sha"]), passes=reps,
                   first_stage_that_varies=first_bad, per_stage=per_stage)
        # the stage checksums of a pass whose targets equal the fixture's: the per-stage reference a differing box is compared with
        good = next((i for i, h in enumerate(ysha) if h == m["y_sha"]), None)
        if good is not None:
            res["stage_checksums_of_a_fixture_matching_pass"] = {s: [int(v) for v in table[s][good]] for s in STAGES}
        out[name] = res
        print(name, json.dumps({k: v for k, v in res.items() if k not in ("per_stage", "stage_checksums_of_a_fixture_matching_pass")}), flush=True)
        print("   ", {s: per_stage[s]["calls_off_majority"] for s in STAGES}, flush=True)

    variant("det_materialised_async")
    variant("det_materialised_sync", sync=True, reps=max(3, R // 3))
    variant("nodet_materialised_async", det=False, reps=max(3, R // 2))
    variant("det_broadcast_async", materialise=False, reps=max(3, R // 3))
    for be in ("efficient", "flash", "math"):
        try:
            variant(f"det_materialised_async_{be}", backend=be, reps=max(3, R // 3))
        except Exception as e:  # noqa: BLE001
            out[f"det_materialised_async_{be}"] = f"failed: {e!r}"[:300]
    del model, block, x0
    torch.cuda.empty_cache()

    runs = []
    for t in range(T):
        r = fx.check_against_stat_fixture(fixp)
        runs.append({k: r[k] for k in ("bit_identical", "targets_identical", "inputs_identical", "tensors_identical", "tensors", "first_divergence_iter",
                                       "prefix_identical_codes", "best_loss_ratio", "result_digest", "tune_s")})
        print("tune", t, runs[-1], flush=True)
    out["tunes"] = runs
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06", "opt_determinism.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


if __name__ == "__main__":
    main()
