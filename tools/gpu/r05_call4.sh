#!/bin/bash
# round 5, GPU call 4: two-run fixture tests under the reference's deterministic-algorithms mode, NT / grouped GEMM tests, the
# wrong-cut-table test, Mixtral block bench with grouped expert GEMMs vs the per-expert loop
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_t3_fixture.py -q -k "two_run or real_width or wrong_stream" > $O/t3s_fixture2.log 2>&1; echo "t3s rc=$?"; tail -30 $O/t3s_fixture2.log | cut -c1-1500
timeout 300 python -m pytest tests/test_gpu_gemm_nt.py -q > $O/gemm_nt_tests.log 2>&1; echo "nt tests rc=$?"; tail -6 $O/gemm_nt_tests.log | cut -c1-600
for g in 1 0; do
  AR_MOE_GROUPED=$g timeout 600 python bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --path fused --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/bench_mixtral_mxfp4_grouped$g.json 2> $O/bench_mixtral_mxfp4_grouped$g.err
  echo "mixtral grouped=$g rc=$?"; python -c "
import json,sys
d=json.loads(open('$O/bench_mixtral_mxfp4_grouped$g.json').read().strip().splitlines()[-1])
print({k:d[k] for k in ('value','ms_per_step','ms_per_iter')}, d['config']['fused_block'], d['loss'])"
done
