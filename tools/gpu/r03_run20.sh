# experiment: does torch's TunableOp find faster library GEMMs for the forward / dX shapes than hipBLASLt's default heuristic?
set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3s; mkdir -p $O
export PYTORCH_TUNABLEOP_ENABLED=1 PYTORCH_TUNABLEOP_TUNING=1 PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=40 PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS=5 PYTORCH_TUNABLEOP_VERBOSE=0
export PYTORCH_TUNABLEOP_FILENAME=$O/tunableop_llama8b.csv
timeout 700 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-kernel-timing --no-hip-graph > $O/b_llama_tuned.json 2> $O/b_llama_tuned.err; echo rc=$?
tail -c 300 $O/b_llama_tuned.err
export PYTORCH_TUNABLEOP_FILENAME=$O/tunableop_opt125m.csv
timeout 400 python bench.py --workload opt-125m --steps 3 --warmup 1 --no-extras --no-cpu-baseline --no-kernel-timing --no-hip-graph > $O/b_opt_tuned.json 2> $O/b_opt_tuned.err; echo rc=$?
unset PYTORCH_TUNABLEOP_ENABLED PYTORCH_TUNABLEOP_TUNING
timeout 400 python bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --no-kernel-timing --no-hip-graph > $O/b_llama_plain.json 2> /dev/null
for f in b_llama_tuned b_opt_tuned b_llama_plain; do python - $O/$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(sys.argv[1], d["value"], d["ms_per_iter"])
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
ls -la $O; head -30 $O/tunableop_llama8b*.csv | cut -c1-200
