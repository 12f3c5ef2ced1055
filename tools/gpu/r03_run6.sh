set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3f; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py -q -m gpu > $O/t_kern.log 2>&1; echo "kern rc=$?"; tail -8 $O/t_kern.log
timeout 2400 python tests/t3_baseline_shapes.py --out $O/t3.json --fixture $O/t3_fixture.npz --digest $O/t3_digest.npz --ref-twice opt125m_w4g128,llama8b_w4g128 --cases opt125m_w4g128,llama8b_w4g128,llama8b_w4g128_full,llama8b_w2g32_asym_algext,llama8b_mxfp4,llama8b_nvfp4 > $O/t3.log 2>&1; echo "t3 rc=$?"
python - <<'PY'
import json
t=json.load(open('gpurun_out/r3f/t3.json'))
for c in t['cases']:
    print(c['case'], c.get('error'), c.get('ref_wall_s'), c.get('ref_vs_ref'))
    print('  probe', {k:v for k,v in (c.get('grad_sign_probe') or {}).items() if 'per_iter' not in k})
    for tag in ('module','fused','alone_module','alone_fused'):
        r=c.get(tag) or {}
        print('  ',tag, {k:r.get(k) for k in ('first_divergence_iter','identical_codes','identical_weights','identical_scale_zp_where_codes_agree','best_loss_ratio','init_loss_rel_diff','hip_graph','targets_identical')})
PY
ls -la $O
