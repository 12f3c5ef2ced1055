# OPT-125M: captured-graph vs host-driven iterations with the first-party attention backward; rocprofv3 kernel stats of the graph form
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3q; mkdir -p $O
for v in "--hip-graph" "--no-hip-graph" "--hip-graph --no-attn-bwd"; do
  n=$(echo $v | tr -d ' ')
  timeout 200 python bench.py --workload opt-125m --steps 5 --warmup 1 --no-kernel-timing --no-cpu-baseline --no-extras $v > "$O/b$n.json" 2> $O/b.err
  python - "$O/b$n.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_iter"], d["config"].get("hip_graph"), d["config"].get("flash_attention_bwd"))
PY
done
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/prof_opt -- python $GRAFT_REPO_ROOT/bench.py --workload opt-125m --steps 3 --warmup 1 --hip-graph --no-extras --no-cpu-baseline --no-kernel-timing > $O/bench_opt125m_under_rocprof.json 2> $O/rocprof.err; echo "rocprof rc=$?"
python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $(find /tmp/prof_opt -name "*.db" | head -1) --stats $O/opt125m_graph_kernel_stats.csv
head -24 $O/opt125m_graph_kernel_stats.csv | cut -c1-160
