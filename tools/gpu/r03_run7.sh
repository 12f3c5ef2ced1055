set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3g; mkdir -p $O
python tests/asym_grad_probe.py 2 32 > $O/asym_w2g32.json 2> $O/asym.err; cat $O/asym_w2g32.json; tail -2 $O/asym.err
python tests/asym_grad_probe.py 4 128 > $O/asym_w4g128.json 2>> $O/asym.err; cat $O/asym_w4g128.json
python tools/gpu/r03_exp_attn_strided.py > $O/attn_strided.json 2> $O/attn_strided.err; cat $O/attn_strided.json; tail -2 $O/attn_strided.err
timeout 900 python -m pytest tests/test_gpu_t3_fixture.py -q -m gpu > $O/t_fix.log 2>&1; echo "fixture rc=$?"; tail -12 $O/t_fix.log
timeout 600 python bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/mixtral_mxfp4_fused_tn.json 2> $O/mixtral_mxfp4_fused_tn.err; echo "mix rc=$?"; tail -c 300 $O/mixtral_mxfp4_fused_tn.err; head -c 400 $O/mixtral_mxfp4_fused_tn.json
timeout 1200 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 500 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3g/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_iter'])
o=d['opt125m']; print({k:o[k] for k in ('value','ms_per_iter','hip_graph')})
print(json.dumps(d['parity'])[:3000])
PY
