set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3t; mkdir -p $O
timeout 200 python tools/gpu/r03_exp_moe_gemm_split.py > $O/moe_gemm_split.jsonl 2> $O/err.log; cat $O/moe_gemm_split.jsonl; tail -2 $O/err.log
