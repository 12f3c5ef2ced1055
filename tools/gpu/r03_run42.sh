set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
for v in "--no-kernel-timing" "--no-kernel-timing --hip-graph" ""; do
  n=$(echo $v | tr -d ' ')
  timeout 200 python bench.py --workload opt-125m --steps 3 --warmup 1 --no-cpu-baseline --no-extras $v > "$O/opt$n.json" 2> $O/b.err
  python - "$O/opt$n.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r=d.get("roofline",{}); b=d.get("roofline_bwd_sgd",{})
print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_iter"], d["config"].get("hip_graph"), d["config"].get("flash_attention_bwd"), r.get("avg_launch_ms"), b.get("avg_launch_ms"), d.get("gemm_dw",{}).get("avg_launch_ms"))
PY
done
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk" | head -4
