#!/bin/bash
# round 5, GPU call 2: NT / grouped GEMM probe, then the digest / fixture tests and the exact-block tests on the changed proof
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/gpu/r05_gemm_nt_probe.py --out $O/gemm_nt_probe.json > $O/gemm_nt_probe.log 2>&1; echo "probe rc=$?"; grep -v amdgpu.ids $O/gemm_nt_probe.log | cut -c1-700
timeout 300 python -m pytest tests/test_gpu_exact_block.py -q -x > $O/exact_block.log 2>&1; echo "exact rc=$?"; tail -15 $O/exact_block.log | cut -c1-600
timeout 900 python -m pytest tests/test_gpu_t3_fixture.py -q > $O/t3_fixture.log 2>&1; echo "t3 fixture rc=$?"; tail -40 $O/t3_fixture.log | cut -c1-900
