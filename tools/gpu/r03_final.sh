# round-3 measurement run: the driver-style bench line, rocprofv3 kernel stats of the same command, PMC traffic passes, MoE kernel stats
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3z; mkdir -p $O
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 400 $O/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_llama -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof llama rc=$?"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_llama -name "*.db" | head -1) --stats $O/llama8b_fused_kernel_stats.csv
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $C -d /tmp/pmc_$C -- python $GRAFT_REPO_ROOT/tools/pmc_traffic_probe.py > $O/pmc_$C.log 2>&1; echo "pmc $C rc=$?"
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/pmc_$C -name "*.db" | head -1) --pmc-rows $O/pmc_rows_$C.csv | tail -1
done
python $GRAFT_REPO_ROOT/tools/pmc_traffic_merge.py $O/pmc_rows_FETCH_SIZE.csv $O/pmc_rows_WRITE_SIZE.csv > $O/pmc_traffic.json; cat $O/pmc_traffic.json | head -30
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_mix -- python $GRAFT_REPO_ROOT/bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 1 --warmup 0 --iters 30 --no-extras --no-cpu-baseline --no-kernel-timing > $O/mixtral_under_rocprof.json 2> $O/mixtral_under_rocprof.err; echo "rocprof mix rc=$?"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_mix -name "*.db" | head -1) --stats $O/mixtral_mxfp4_fused_kernel_stats_30iters.csv
head -30 $O/mixtral_mxfp4_fused_kernel_stats_30iters.csv | cut -c1-180
head -16 $O/llama8b_fused_kernel_stats.csv | cut -c1-180
