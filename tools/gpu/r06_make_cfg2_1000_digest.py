"""BASELINE configs[2] at ITS OWN LENGTH: Llama-3-8B block dimensions, W2 group_size=32 asym, the algorithm
extension, iters = 1000 (so lr = 2 / iters by the reference's rule for <= 3 bits, auto_round/algorithms/quantization/sign_round/
config.py:110-140), 64 x 2048 calibration tokens, batch 8, seed 42 -- 125 passes over the samples, |V| free to reach 1.0.

The reference cannot run on the GPU box (its tree does not travel), so this digest is NOT reference-made: it is made by THIS package's
MODULE PATH -- transformers' module code around the quant kernels, the path that reproduces the reference-made 200-iteration digest of
the same configuration bit for bit (tests/golden/t3v2_llama8b_w2g32_asym_algext_200.npz) -- run twice here; both runs must agree before
the file is written, and the file says who made it.  What it pins in the driver's suite: that `exact_rounding` (the headline's path) stays
bit-identical to the module path over a 5x longer, chaotic trajectory (tests/test_gpu_t3_fixture.py), and the time per block.

    python tools/gpu/r06_make_cfg2_1000_digest.py   ->  gpurun_out/r06/t3m_llama8b_w2g32_asym_algext_1000.npz (+ .json report)
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd.testing import t3_fixture as fx  # noqa: E402

CASE = dict(arch="llama8b", scheme="W2A16G32", kw=dict(sym=False, enable_alg_ext=True), iters=1000, nsamples=64, seqlen=2048, batch_size=8)


def run(exact):
    t0 = time.perf_counter()
    r = fx.tune_with_product(CASE["arch"], scheme=CASE["scheme"], scheme_kw=dict(sym=False), iters=CASE["iters"], nsamples=CASE["nsamples"],
                             seqlen=CASE["seqlen"], batch_size=CASE["batch_size"], alg_ext=True, exact=exact, seed=42)
    return r, fx.tuned_layer_tensors(r["block"]), time.perf_counter() - t0


def main():
    out_dir = os.path.join(ROOT, "gpurun_out", "r06")
    os.makedirs(out_dir, exist_ok=True)
    r1, t1, w1 = run(False)
    d1 = fx.digest_of(t1)
    r2, t2, w2 = run(False)
    d2 = fx.digest_of(t2)
    re_, te, we = run(True)
    de = fx.digest_of(te)
    rep = {"case": CASE, "torch": torch.__version__, "device": torch.cuda.get_device_name(0),
           "module_run1_vs_run2_identical": d1 == d2, "exact_vs_module_identical": de == d1, "exact_block": bool(re_["exact_block"]),
           "exact_plan": (re_["exact_report"] or {}).get("plan"), "lr": None,
           "module_tune_s": [r1["tune_s"], r2["tune_s"]], "exact_tune_s": re_["tune_s"], "wall_s": [w1, w2, we],
           "best_iter": [r1["stats"]["best_iter"], re_["stats"]["best_iter"]], "init_loss": r1["stats"]["init_loss"],
           "best_loss": [r1["stats"]["best_loss"], r2["stats"]["best_loss"], re_["stats"]["best_loss"]],
           "first_divergence_module_runs": fx.trace_divergence(r1["loss_trace"], r2["loss_trace"]),
           "first_divergence_exact_vs_module": fx.trace_divergence(r1["loss_trace"], re_["loss_trace"])}
    print(json.dumps(rep), flush=True)
    if d1 == d2:
        kw = dict(CASE["kw"])
        path = os.path.join(out_dir, "t3m_llama8b_w2g32_asym_algext_1000.npz")
        case = dict(CASE, kw=kw)
        fx.write_digest_v2(path, case, t1, r1["loss_trace"], r1["x_sha"], r1["y_sha"],
                           dict(device=torch.cuda.get_device_name(0), torch=torch.__version__,
                                made_by="tools/gpu/r06_make_cfg2_1000_digest.py: THIS package's module path (two identical runs), NOT the reference -- "
                                        "the path that reproduces the reference-made 200-iteration digest of the same configuration bit for bit"))
        rep["digest_bytes"] = os.path.getsize(path)
    with open(os.path.join(out_dir, "cfg2_1000_digest_report.json"), "w") as f:
        json.dump(rep, f, indent=1, default=str)


if __name__ == "__main__":
    main()
