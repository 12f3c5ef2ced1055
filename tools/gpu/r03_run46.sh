set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
timeout 100 python -m pytest tests/test_gpu_t3_fixture.py -q -m gpu -x -k "fused_path_stays_on_the_reference_trajectory_level and llama8b" 2>&1 | tail -12 | cut -c1-1500
