set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused_block.py -q -m gpu -k "opt or OPT or graph or attention" -x > $O/t.log 2>&1; echo "rc=$?"; tail -8 $O/t.log
for v in "" "--no-attn-bwd" "--no-hip-graph" "--no-hip-graph --no-attn-bwd"; do
  timeout 200 python bench.py --workload opt-125m --steps 3 --warmup 1 $v > "$O/b$(echo $v | tr -d ' ').json" 2> $O/b.err; tail -2 $O/b.err
  python - "$O/b$(echo $v | tr -d ' ').json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["config"].get("hip_graph"), d["config"].get("flash_attention_bwd"), d.get("roofline",{}).get("frac"), d.get("parity"))
PY
done
