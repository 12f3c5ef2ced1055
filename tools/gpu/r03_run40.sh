set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
timeout 300 python tools/gpu/r03_gemm_fwd_ab.py > $O/gemm_fwd_vs_hipblaslt.jsonl 2> $O/gemm_fwd.err; echo rc=$?; cat $O/gemm_fwd_vs_hipblaslt.jsonl; tail -3 $O/gemm_fwd.err
