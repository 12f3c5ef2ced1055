#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O
timeout 300 python tools/gpu/r05_gemm_nt_trace.py --out $O/gemm_nt_phase_cycles.json > $O/gemm_nt_trace.log 2>&1; echo "trace rc=$?"; grep -v amdgpu.ids $O/gemm_nt_trace.log | cut -c1-1400
timeout 900 python tools/gpu/r05_moe_target_diag.py --out $O/moe_target_diag.json > $O/moe_target_diag.log 2>&1; echo "diag rc=$?"; grep "^det_\|Error\|error" $O/moe_target_diag.log | cut -c1-3000; tail -3 $O/moe_target_diag.log | cut -c1-600
