"""How often does a 200-iteration run part from a reference-made digest that the same code reproduces at other times?  Repeats the
module path and the exact_rounding path on one digest in ONE process (what the test suite does) and prints each verdict."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd.testing import t3_fixture as fx  # noqa: E402

path = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/t3v2_llama8b_mxfp4_200.npz"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
out = []
for i in range(n):
    for exact in (False, True):
        r = fx.check_against_digest_v2(path, exact=exact)
        rec = dict(run=i, exact=exact, bit_identical=r["bit_identical"], tensors_identical=r["tensors_identical"],
                   first_divergence_iter=r["first_divergence_iter"], best_loss_ratio=r["best_loss_ratio"], tune_s=round(r["tune_s"], 2))
        out.append(rec)
        print(json.dumps(rec), flush=True)
os.makedirs("gpurun_out/r04g", exist_ok=True)
json.dump(dict(digest=os.path.basename(path), runs=out), open("gpurun_out/r04g/digest_repeat.json", "w"), indent=1)
