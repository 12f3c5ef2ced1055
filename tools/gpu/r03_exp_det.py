"""experiment: run-to-run determinism of this package's module path at the OPT-125M BASELINE shape, 80 iterations, three runs in one
process: `plain` (library weight-gradient GEMM) or `mfma` (this repository's deterministic one).  -> profiles/r03_opt125m_determinism.json
(the committed file also holds a third mode, the library GEMM into a fresh tensor, whose switch was removed again)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from auto_round_amd.testing import t3_fixture as fx
import auto_round_amd.quantizer as Q

mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
orig = Q.SignRoundConfig.__init__
if mode == "mfma":
    def patched(self, *a, **k):
        orig(self, *a, **k)
        self.mfma_dw_gemm = True
    Q.SignRoundConfig.__init__ = patched
runs = []
for i in range(3):
    r = fx.tune_with_product("opt125m", fused=False, iters=80)
    runs.append((r["loss_trace"], r["y_sha"]))
first = [next((i for i, (x, y) in enumerate(zip(runs[0][0], runs[j][0])) if x != y), None) for j in (1, 2)]
print(json.dumps({"mode": mode, "AR_DW_VIA_TEMP": os.environ.get("AR_DW_VIA_TEMP"), "first_different_iteration_vs_run0": first,
                  "same_targets": [runs[0][1] == runs[j][1] for j in (1, 2)]}))
