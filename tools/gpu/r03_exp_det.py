"""experiment: is the module path run-to-run deterministic at the OPT-125M BASELINE shape (library GEMMs incl. stream-K ones)?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from auto_round_amd.testing import t3_fixture as fx
runs = []
for i in range(2):
    r = fx.tune_with_product("opt125m", fused=False)
    runs.append((r["loss_trace"], {n: m.weight.detach().clone() for n, m in r["block"].named_modules() if isinstance(m, torch.nn.Linear)}, r["y_sha"]))
a, b = runs
same_trace = a[0] == b[0]
first = next((i for i, (x, y) in enumerate(zip(a[0], b[0])) if x != y), None)
same_w = all(torch.equal(a[1][n], b[1][n]) for n in a[1])
print(json.dumps({"same_loss_trace": same_trace, "first_different_iteration": first, "same_weights": same_w, "same_targets": a[2] == b[2]}))
