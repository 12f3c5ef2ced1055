set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3l; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_autograd_bits.py tests/test_gpu_kernels.py tests/test_gpu_random_ranges.py -q -m gpu > $O/t.log 2>&1; echo "rc=$?"; tail -30 $O/t.log
