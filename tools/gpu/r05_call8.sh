#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_t3_fixture.py -q -k "two_run or real_width or fixture_is or module_path_reproduces_the_reference_on_gpu_fixture or fused_path_stays" > $O/t3s_fixture3.log 2>&1; echo "t3s rc=$?"; tail -30 $O/t3s_fixture3.log | cut -c1-1800
