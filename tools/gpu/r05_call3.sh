#!/bin/bash
# round 5, GPU call 3: A/B of the DMA-issue variants (NT + weight-gradient kernels), grouped GEMMs with the banded tile order,
# the two-run fixture tests (module + fused MoE), the grouped-vs-loop MoE test, the PMC traffic passes of K1 / K2
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O
timeout 600 python tools/gpu/r05_gemm_nt_probe.py --ab --out $O/gemm_ab_probe.json > $O/gemm_ab_probe.log 2>&1; echo "ab probe rc=$?"; grep -v amdgpu.ids $O/gemm_ab_probe.log | cut -c1-900
timeout 300 python -m pytest tests/test_gpu_fused_block.py -q -x -k "moe or MoE or grouped" > $O/moe_tests.log 2>&1; echo "moe rc=$?"; tail -8 $O/moe_tests.log | cut -c1-700
timeout 900 python -m pytest tests/test_gpu_t3_fixture.py -q -k "two_run or real_width" > $O/t3s_fixture.log 2>&1; echo "t3s rc=$?"; tail -30 $O/t3s_fixture.log | cut -c1-1200
bash tools/gpu/r05_pmc_traffic.sh 2>&1 | tail -25 | cut -c1-400
