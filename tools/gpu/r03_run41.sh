set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=$GRAFT_REPO_ROOT/gpurun_out/r3x; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_t3_fixture.py tests/test_gpu_fused_block.py -q -m gpu -x -k "fixture or graph or reproducible or fused_path or module_path or digest" > $O/t2.log 2>&1; echo "rc=$?"; tail -4 $O/t2.log
timeout 1500 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python - $O/bench_default.json <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
o=d["opt125m"]
print(d["value"], d["ms_per_iter"], d["roofline"]["frac"], d["roofline_bwd_sgd"]["frac"])
print("opt", o["value"], o["ms_per_iter"], o["hip_graph"], o["other_launch_form"], o["host_driven_block"]["ms_per_iter"])
print(d["parity"]["llama8b_module_path_bit_identical"], d["parity"]["module_path_identical_codes"], d["parity"]["fused_path_identical_codes"], d["parity"]["fused_path"]["hip_graph"])
PY
