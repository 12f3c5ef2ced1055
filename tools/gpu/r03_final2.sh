# round-3 closing run: the whole -m gpu suite, the driver-style bench line, rocprofv3 kernel stats of the same command
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3y; mkdir -p $O
timeout 1200 python -m pytest tests -q -m gpu -x > $O/gpu_suite.log 2>&1; echo "suite rc=$?"; tail -4 $O/gpu_suite.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"; tail -c 300 $O/bench_default.err
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/prof_llama -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err; echo "rocprof llama rc=$?"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_llama -name "*.db" | head -1) --stats $O/llama8b_fused_kernel_stats.csv
head -8 $O/llama8b_fused_kernel_stats.csv | cut -c1-160
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_step"], d.get("ms_per_iter"), d["roofline"]["frac"], d.get("roofline_bwd_sgd",{}).get("frac"))
print(json.dumps(d.get("parity"))[:1500])
print(json.dumps(d.get("opt125m"))[:1800])
print(json.dumps(d.get("cpu_baseline"))[:600])
PY
