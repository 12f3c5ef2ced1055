set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
export OUT=$GRAFT_REPO_ROOT/gpurun_out/r3w; mkdir -p $OUT
NO_MASK=1 ITERS=1500 timeout 300 python tools/gpu/r03_exp_det7.py > $OUT/log_nomask.txt 2>&1; echo rc=$?; grep -v "amdgpu.ids\|layer_idx" $OUT/log_nomask.txt | tail -8 | cut -c1-400
for i in 1 2 3; do timeout 300 python -m pytest tests/test_gpu_fused_block.py -q -m gpu -k "reproducible or attention_backward" -x 2>&1 | tail -2; done
