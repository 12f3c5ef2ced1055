"""Repeat the reference-free Mixtral module-path flow N times per fixture (optionally under another attention launch form:
AR_XATTN_CFG = ar_attn_exact_config bits) and report, per run, whether the reference-made fixture was reproduced bit for bit."""
import glob, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd.testing import t3_fixture as fx
from auto_round_amd import _lib
root = os.environ.get("GRAFT_REPO_ROOT", ".")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 6
which = sys.argv[2] if len(sys.argv) > 2 else "mixtral"
cfgs = [int(x) for x in os.environ.get("AR_XATTN_CFG", "0").split(",")]          # several forms: alternated run by run
lib = _lib.load()
res = []
paths = sorted(glob.glob(os.path.join(root, "tests", "golden", f"t3s_{which}*.npz")))
interleave = os.environ.get("AR_INTERLEAVE") == "1"          # fixture by fixture inside every round: a MIXED process (other blocks in between)
order = [(i, p) for i in range(N) for p in paths] if interleave else [(i, p) for p in paths for i in range(N)]
for i, path in order:
    if True:
        cfg = cfgs[i % len(cfgs)]
        lib.ar_attn_exact_config(cfg)
        r = fx.check_against_stat_fixture(path, exact=("opt125m" in path))
        rec = {"fixture": os.path.basename(path), "run": i, "attn_cfg": cfg, **{k: r[k] for k in ("targets_identical", "bit_identical", "tensors_identical", "first_divergence_iter", "prefix_identical_weights")}}
        res.append(rec)
        print(json.dumps(rec), flush=True)
out = os.path.join(root, "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
lib.ar_attn_exact_config(0)
summary = {c: {"runs": sum(1 for r in res if r["attn_cfg"] == c), "bit_identical": sum(1 for r in res if r["attn_cfg"] == c and r["bit_identical"])} for c in cfgs}
print("SUMMARY", json.dumps(summary), flush=True)
json.dump({"runs": res, "summary": summary}, open(os.path.join(out, "mixtral_fixture_repeat_cfg" + "_".join(str(c) for c in cfgs) + ("_interleaved" if interleave else "") + ".json"), "w"), indent=1)
