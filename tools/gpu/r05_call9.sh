#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 AR_T3_KEEP_TARGETS=1
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/gpu/r05_gemm_nt_probe.py --ab --out $O/gemm_ab_probe2.json > $O/gemm_ab_probe2.log 2>&1; echo "ab probe rc=$?"; grep -v amdgpu.ids $O/gemm_ab_probe2.log | head -3 | cut -c1-900; grep '"counts"' $O/gemm_ab_probe2.log | cut -c1-700
timeout 200 python -m pytest tests/test_gpu_gemm_nt.py -q > $O/gemm_nt_tests2.log 2>&1; echo "nt tests rc=$?"; tail -5 $O/gemm_nt_tests2.log | cut -c1-500
timeout 900 python tests/t3_baseline_shapes.py --cases mixtral8x7b_mxfp4_2 --variants module --out $O/t3_mixtral_targets_diag2.json > $O/t3_mixtral_targets_diag2.log 2>&1
echo "rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/t3_mixtral_targets_diag2.json'))
c=d['cases'][0]
print('err', c.get('error'), (c.get('trace') or '')[-2500:])
fc=c.get('forward_compare') or {}
for k in ('flags_ref','flags_mine','attn_kwargs_ref','attn_kwargs_mine'): print(k, json.dumps(fc.get(k))[:1500])
print(json.dumps(fc.get('outputs'), indent=0)[:4000])
print('alone targets', (c.get('alone_module') or {}).get('targets_compare'))
PY
