set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -6 $O/gpu_suite.log
