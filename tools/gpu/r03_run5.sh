set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3e; mkdir -p $O
python tools/gpu/r03_exp_det.py > $O/det.json 2> $O/det.err; echo "det rc=$?"; cat $O/det.json; tail -2 $O/det.err
timeout 1500 python tests/t3_baseline_shapes.py --out $O/t3_full.json --cases llama8b_w4g128_full --skip-alone > $O/t3_full.log 2>&1; echo "t3 rc=$?"
python - <<'PY'
import json
t=json.load(open('gpurun_out/r3e/t3_full.json'))
for c in t['cases']:
    print(c['case'], c.get('error'), c.get('ref_wall_s'))
    print('  probe', {k:v for k,v in (c.get('grad_sign_probe') or {}).items() if 'per_iter' not in k})
    for tag in ('module','fused'):
        r=c.get(tag) or {}
        print('  ',tag, {k:r.get(k) for k in ('first_divergence_iter','identical_codes','identical_weights','identical_scale_zp_where_codes_agree','best_loss_ratio','init_loss_rel_diff','hip_graph','wall_s')})
PY
