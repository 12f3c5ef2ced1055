set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3o; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_fused_block.py -q -m gpu -k "attention_backward" -x > $O/t.log 2>&1; echo "rc=$?"; tail -25 $O/t.log
timeout 120 python tools/gpu/r03_attn_bwd_probe.py > $O/probe.json 2> $O/probe.err; cat $O/probe.json; tail -3 $O/probe.err
