"""Round 6: how often does a full-recipe run reproduce the reference-made fixture, run after run, in one process?

  OPT-125M (tests/golden/t3s_opt125m_w4g128.npz; 200 iterations, 128 x 2048, batch 8), N runs each of
      module path / exact_rounding / exact_rounding + verify_attention_forward / module path with the library's inference-mode
      attention for the no-grad forwards (what the reference itself runs; rounds 3-5)
  Llama-3-8B W4G128 (the headline's digest), M runs on exact_rounding.

    python tools/gpu/r06_parity_repeat.py [N] [M]  ->  gpurun_out/r06/parity_repeat.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd.testing import t3_fixture as fx  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 30
    M = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    out = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "opt_runs_per_variant": N, "llama_runs": M}
    fixp = os.path.join(ROOT, "tests", "golden", "t3s_opt125m_w4g128.npz")
    variants = (("opt125m_module_path", dict()), ("opt125m_exact_rounding", dict(exact=True)),
                ("opt125m_exact_rounding_verified", dict(exact=True, verify_attention=True)),
                ("opt125m_module_path_inference_mode_forwards", dict(reproducible_attention=False)))
    for name, kw in (variants if N > 0 else ()):
        runs = []
        for _ in range(N):
            r = fx.check_against_stat_fixture(fixp, **kw)
            runs.append({k: r.get(k) for k in ("bit_identical", "targets_identical", "first_divergence_iter", "first_differing_stage",
                                               "attention_forward_retries", "exact_block", "tune_s")})
        rec = {"runs": N, "bit_identical": sum(bool(r["bit_identical"]) for r in runs), "targets_identical": sum(bool(r["targets_identical"]) for r in runs),
               "retries": sum(int(r["attention_forward_retries"] or 0) for r in runs), "median_tune_s": sorted(r["tune_s"] for r in runs)[N // 2],
               "parted_at": [r["first_divergence_iter"] for r in runs if not r["bit_identical"]],
               "stages": [r["first_differing_stage"] for r in runs if not r["targets_identical"]]}
        out[name] = rec
        print(name, rec, flush=True)
    runs = []
    for _ in range(M):
        r = fx.check_against_digest(exact=True)
        runs.append({k: r.get(k) for k in ("bit_identical", "targets_identical", "first_divergence_iter", "tune_s", "exact_block")})
    out["llama8b_exact_rounding"] = {"runs": M, "bit_identical": sum(bool(r["bit_identical"]) for r in runs),
                                     "targets_identical": sum(bool(r["targets_identical"]) for r in runs),
                                     "median_tune_s": sorted(r["tune_s"] for r in runs)[M // 2],
                                     "parted_at": [r["first_divergence_iter"] for r in runs if not r["bit_identical"]]}
    print("llama8b_exact_rounding", out["llama8b_exact_rounding"], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06", "parity_repeat.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


if __name__ == "__main__":
    main()
