set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_t3_fixture.py -q -m gpu -k "reproducible or fixture or digest or fused_path or module_path" -x > $O/t.log 2>&1; echo "rc=$?"; tail -30 $O/t.log
