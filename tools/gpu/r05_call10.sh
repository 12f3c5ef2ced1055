#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_t3_fixture.py -q -k "two_run or real_width" > $O/t3s_fixture4.log 2>&1; echo "t3s rc=$?"; tail -40 $O/t3s_fixture4.log | cut -c1-2500
