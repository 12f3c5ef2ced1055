set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w; mkdir -p $OUT
timeout 500 python tools/gpu/r03_exp_det9.py > $OUT/log.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids\|layer_idx" $OUT/log.txt | tail -60
