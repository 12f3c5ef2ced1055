#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 AR_T3_KEEP_TARGETS=1
O=gpurun_out/r05; mkdir -p $O
timeout 900 python tests/t3_baseline_shapes.py --cases mixtral8x7b_mxfp4_2 --variants module --out $O/t3_mixtral_targets_diag.json > $O/t3_mixtral_targets_diag.log 2>&1
echo "rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/t3_mixtral_targets_diag.json'))
c=d['cases'][0]
print('err', c.get('error'), (c.get('trace') or '')[-1500:])
print('ref inputs', {k:c['ref']['inputs'].get(k) for k in ('y_dtype','y_shape','x_sha','y_sha')})
for k in ('module','alone_module'):
    if k in c:
        r=c[k]; print(k, {kk: r.get(kk) for kk in ('targets_compare','inputs_identical','targets_identical','first_divergence_iter','identical_weights','init_loss','init_loss_rel_diff')}, (r.get('stats') or {}).get('init_loss'))
PY
