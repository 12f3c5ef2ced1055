"""Round 6 A/B: does the launch sequence around the library attention call change how often a full-recipe OPT-125M run parts from the
reference-made fixture?  N runs of `check_against_stat_fixture` per (path, guard) pair, guards from attention.guarded_sdpa.

    python tools/gpu/r06_sdpa_guard_ab.py [N]   ->  gpurun_out/r06/sdpa_guard_ab.json
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd.testing import t3_fixture as fx  # noqa: E402


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 24
    fixp = os.path.join(ROOT, "tests", "golden", "t3s_opt125m_w4g128.npz")
    out = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "runs": N}
    guards = sys.argv[2].split("/") if len(sys.argv) > 2 else ["", "before", "after", "before,after", "touch"]
    paths = (True, False) if len(sys.argv) <= 3 else tuple(p == "exact" for p in sys.argv[3].split("/"))
    for exact in paths:
        for guard in guards:
            os.environ["AR_SDPA_GUARD"] = guard
            runs = [fx.check_against_stat_fixture(fixp, exact=exact) for _ in range(N)]
            rec = {"bit_identical": sum(bool(r["bit_identical"]) for r in runs), "targets_identical": sum(bool(r["targets_identical"]) for r in runs),
                   "median_tune_s": sorted(r["tune_s"] for r in runs)[N // 2], "parted_at": [r["first_divergence_iter"] for r in runs if not r["bit_identical"]]}
            out[f"{'exact' if exact else 'module'}|{guard or 'none'}"] = rec
            print(f"{'exact' if exact else 'module'}|{guard or 'none'}", rec, flush=True)
    os.environ.pop("AR_SDPA_GUARD", None)
    tag = "" if len(sys.argv) <= 2 else "_" + str(abs(hash(sys.argv[2])) % 1000)
    with open(os.path.join(ROOT, "gpurun_out", "r06", f"sdpa_guard_ab{tag}.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
