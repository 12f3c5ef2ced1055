set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_fused_block.py tests/test_gpu_bench.py -q -m gpu -x > $O/t.log 2>&1; echo "rc=$?"; tail -5 $O/t.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["value"], d["ms_per_iter"], d["roofline"]["frac"], d["roofline_bwd_sgd"]["frac"], d["opt125m"]["value"], d["opt125m"]["ms_per_iter"], d["opt125m"]["host_driven_block"]["ms_per_iter"])
print(d["parity"]["llama8b_module_path_bit_identical"], d["parity"]["module_path_identical_codes"], d["parity"]["fused_path_identical_codes"], d["config"]["attention_mask"][:30])
PY
