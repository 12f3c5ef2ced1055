set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3c; mkdir -p $O
python tools/gpu/r03_exp_mask.py > $O/exp_mask.jsonl 2> $O/exp_mask.err; echo "exp rc=$?"; cat $O/exp_mask.jsonl; tail -3 $O/exp_mask.err
timeout 600 python bench.py --workload opt-125m --no-cpu-baseline --no-extras --no-kernel-timing --steps 6 --warmup 2 > $O/opt_graph.json 2> $O/opt_graph.err; echo "rc=$?"; tail -c 300 $O/opt_graph.err; head -c 1600 $O/opt_graph.json
timeout 600 python -m pytest tests/test_gpu_fused_block.py -q -m gpu -k "moe or hipgraph or nvfp4" > $O/t_fused.log 2>&1; echo "fused rc=$?"; tail -5 $O/t_fused.log
timeout 600 python bench.py --workload mixtral-8x7b-hf --scheme NVFP4 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/mixtral_nvfp4_fused.json 2> $O/mixtral_nvfp4_fused.err; echo "mix nv rc=$?"; tail -c 400 $O/mixtral_nvfp4_fused.err; head -c 1500 $O/mixtral_nvfp4_fused.json
