# builder side, reference staged: T3 at the Mixtral-8x7B block (BASELINE configs[4]) -- the real reference on cuda:0 vs the plugin
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
timeout 280 python tests/t3_baseline_shapes.py --cases mixtral_tiny_mxfp4,mixtral8x7b_mxfp4 --skip-alone --out $O/t3_mixtral.json > $O/t3_mixtral.log 2>&1; echo rc=$?
tail -5 $O/t3_mixtral.log | cut -c1-300
python - $O/t3_mixtral.json <<'PY'
import json,sys
try:
    d=json.load(open(sys.argv[1]))
    for r in d["cases"]:
        print(r["case"], "err:", r.get("error"), "ref_wall", r.get("ref_wall_s"))
        for t in ("module","fused"):
            m=r.get(t) or {}
            print("  ",t,{k:m.get(k) for k in ("fused_block","identical_weights","identical_codes","first_divergence_iter","best_loss_ratio","init_loss_rel_diff","wall_s")})
        print("  probe", json.dumps(r.get("grad_sign_probe"))[:300])
except Exception as e: print("no json", e)
PY
