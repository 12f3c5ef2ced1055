set -x
cd /tmp
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_opt -- python $GRAFT_REPO_ROOT/bench.py --workload opt-125m --steps 3 --warmup 1 --hip-graph --no-extras --no-cpu-baseline --no-kernel-timing > $O/bench_opt125m_under_rocprof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_opt -name "*.db" | head -1) --stats $O/opt125m_graph_kernel_stats.csv
python - $O/opt125m_graph_kernel_stats.csv $O/bench_opt125m_under_rocprof.json <<'PY'
import csv,sys,json
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['total_us']) for r in rows)
print("kernel time total ms", tot/1000, "per iteration (800)", tot/800)
for r in rows[:12]: print(r['kernel'][:70], r['calls'], r['avg_us'], r['percent'])
d=json.loads(open(sys.argv[2]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_iter"])
PY
