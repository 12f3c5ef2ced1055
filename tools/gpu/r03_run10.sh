set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3j; mkdir -p $O
timeout 2400 python -m pytest tests/ -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"; tail -15 $O/t_all.log
