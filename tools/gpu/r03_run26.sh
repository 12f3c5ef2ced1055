set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3u; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_fused_block.py -q -m gpu -k "moe or MoE" -x > $O/t.log 2>&1; echo "rc=$?"; tail -5 $O/t.log
timeout 300 python bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 1 --warmup 0 --no-extras --no-cpu-baseline --no-kernel-timing > $O/mix_mxfp4.json 2> $O/mix.err; tail -2 $O/mix.err
python - $O/mix_mxfp4.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(d["value"], d["ms_per_step"], d["ms_per_iter"])
PY
