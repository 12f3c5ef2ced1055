set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "int_bwd or three_fused or midsize or full_size_llama or generic_group or algorithm_extension_wrapper" > $O/t_kern.log 2>&1; echo "kern rc=$?"; tail -8 $O/t_kern.log
timeout 600 python -m pytest tests/test_gpu_fused_block.py -q -m gpu -k "nvfp4" > $O/t_nv.log 2>&1; echo "nv rc=$?"; tail -8 $O/t_nv.log
timeout 1500 python tests/t3_baseline_shapes.py --out $O/t3.json --fixture $O/t3_fixture.npz --skip-alone > $O/t3.log 2>&1; echo "t3 rc=$?"
python - <<'PY'
import json
t=json.load(open('gpurun_out/r3d/t3.json'))
for c in t['cases']:
    print(c['case'], c.get('error'))
    print('  probe', {k:v for k,v in (c.get('grad_sign_probe') or {}).items() if 'per_iter' not in k})
    for tag in ('module','fused'):
        r=c.get(tag) or {}
        print('  ',tag, {k:r.get(k) for k in ('first_divergence_iter','identical_codes','identical_weights','identical_scale_zp_where_codes_agree','best_loss_ratio','init_loss_rel_diff','hip_graph')})
PY
