set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3i; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu -x > $O/t_all.log 2>&1; echo "all rc=$?"; tail -15 $O/t_all.log
timeout 1500 python tests/t3_baseline_shapes.py --out $O/t3.json --cases llama8b_w2g32_asym_algext,llama8b_mxfp4,llama8b_nvfp4 > $O/t3.log 2>&1; echo "t3 rc=$?"
python - <<'PY'
import json
t=json.load(open('gpurun_out/r3i/t3.json'))
for c in t['cases']:
    print(c['case'], c.get('error'), c.get('ref_wall_s'), c.get('ref_vs_ref'))
    print('  probe', {k:v for k,v in (c.get('grad_sign_probe') or {}).items() if 'per_iter' not in k})
    for tag in ('module','fused','alone_module','alone_fused'):
        r=c.get(tag) or {}
        print('  ',tag, {k:r.get(k) for k in ('first_divergence_iter','identical_codes','identical_weights','identical_scale_zp_where_codes_agree','best_loss_ratio','init_loss_rel_diff','hip_graph','targets_identical')})
PY
