"""The two Mixtral fixtures (made by the REAL reference) against the reference-free flow once the capture forward runs where the reference
runs it -- on the CPU: targets, loss trace and every tuned tensor."""
import glob, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
from auto_round_amd.testing import t3_fixture as fx
root = os.environ.get("GRAFT_REPO_ROOT", ".")
res = {}
for path in sorted(glob.glob(os.path.join(root, "tests", "golden", "t3s_mixtral*.npz"))):
    for fused in (False, True):
        r = fx.check_against_stat_fixture(path, fused=fused)
        keep = {k: r[k] for k in ("fused_block", "inputs_identical", "targets_identical", "bit_identical", "tensors", "tensors_identical", "prefix_identical_weights",
                                  "prefix_identical_scales", "init_loss", "init_loss_ref", "best_loss", "best_loss_ref", "best_loss_ratio", "first_divergence_iter", "tune_s")}
        res[os.path.basename(path) + (" fused" if fused else " module")] = keep
        print(os.path.basename(path), "fused" if fused else "module", json.dumps(keep), flush=True)
out = os.path.join(root, "gpurun_out", "r06"); os.makedirs(out, exist_ok=True)
json.dump(res, open(os.path.join(out, "mixtral_fixture_after_cpu_capture.json"), "w"), indent=1)
