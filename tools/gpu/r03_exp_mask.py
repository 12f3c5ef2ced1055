"""experiment: does handing the module path a materialised [batch, 1, S, S] mask (what the reference's input cache produces)
instead of one broadcast row reproduce the reference's targets / delay the first divergence?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd.testing import t3_fixture as fx
for mat in (False, True):
    r = fx.check_against_fixture(fused=False, materialise=mat)
    print(json.dumps({"materialise": mat, **{k: v for k, v in r.items() if k != "per_layer_identical_codes"}}))
