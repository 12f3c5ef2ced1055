set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "fp4 or activation" > $O/t_fp4.log 2>&1; echo "fp4 rc=$?"; tail -6 $O/t_fp4.log
timeout 1500 python tests/t3_baseline_shapes.py --out $O/t3.json --cases llama8b_mxfp4 --skip-alone > $O/t3.log 2>&1; echo "t3 rc=$?"
python - <<'PY'
import json
t=json.load(open('gpurun_out/r3k/t3.json'))
for c in t['cases']:
    print(c['case'], c.get('error'), c.get('ref_wall_s'))
    print('  probe', {k:v for k,v in (c.get('grad_sign_probe') or {}).items() if 'per_iter' not in k})
    for tag in ('module','fused'):
        r=c.get(tag) or {}
        print('  ',tag, {k:r.get(k) for k in ('first_divergence_iter','identical_codes','identical_weights','identical_scale_zp_where_codes_agree','best_loss_ratio','init_loss_rel_diff')})
PY
