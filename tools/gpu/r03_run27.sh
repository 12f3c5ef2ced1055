# round-3 MoE measurements after the round-aligned expert GEMM calls: Mixtral-8x7B block MXFP4 / NVFP4 (fused MoE path), kernel stats of 30 iterations
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3v; mkdir -p $O
timeout 300 python bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-kernel-timing > $O/bench_mixtral_mxfp4_fused.json 2> $O/e1.err
timeout 300 python bench.py --workload mixtral-8x7b-hf --scheme NVFP4 --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-kernel-timing > $O/bench_mixtral_nvfp4_fused.json 2> $O/e2.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_mix -- python $GRAFT_REPO_ROOT/bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 1 --warmup 0 --iters 30 --no-extras --no-cpu-baseline --no-kernel-timing > $O/mixtral_under_rocprof.json 2> $O/mixtral_under_rocprof.err; echo "rocprof mix rc=$?"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_mix -name "*.db" | head -1) --stats $O/mixtral_mxfp4_fused_kernel_stats_30iters.csv
cd $GRAFT_REPO_ROOT
for f in bench_mixtral_mxfp4_fused bench_mixtral_nvfp4_fused; do python - $O/$f.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_step"], d["ms_per_iter"])
PY
done
head -24 $O/mixtral_mxfp4_fused_kernel_stats_30iters.csv | cut -c1-150
