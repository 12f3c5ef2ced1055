#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1 AR_T3_KEEP_TARGETS=1
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/gpu/r05_gemm_nt_trace.py --out $O/gemm_nt_phase_cycles.json > $O/gemm_nt_trace.log 2>&1; echo "trace rc=$?"; grep -v amdgpu.ids $O/gemm_nt_trace.log | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['K'], d['dma_issue'][:20], 'clock', round(d['shader_clock_ghz_during_the_k_loop'],3), 'PF', round(d['pflops_traced'],3), 'pipe busy', round(d['mfma_pipe_busy_fraction_of_a_simd'],3), 'PF if never idle', round(d['pflops_at_this_clock_if_the_pipe_never_idled'],3))
"
timeout 900 python tests/t3_baseline_shapes.py --cases mixtral8x7b_mxfp4_2 --variants module --out $O/t3_mixtral_targets_diag.json > $O/t3_mixtral_targets_diag.log 2>&1
echo "rc=$?"; python - <<'PY'
import json
d=json.load(open('gpurun_out/r05/t3_mixtral_targets_diag.json'))
c=d['cases'][0]
print('err', c.get('error'), (c.get('trace') or '')[-2500:])
print(json.dumps(c.get('forward_compare'), indent=0)[:6000])
PY
