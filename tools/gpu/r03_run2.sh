set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_fused_block.py -q -m gpu > $O/t_fused.log 2>&1; echo "fused rc=$?"
tail -40 $O/t_fused.log
timeout 1500 python tests/t3_baseline_shapes.py --out $O/t3.json --fixture $O/t3_fixture.npz > $O/t3.log 2>&1; echo "t3 rc=$?"
grep -c . $O/t3.log; ls -la $O
cp $O/t3_fixture.npz tests/golden/t3_opt125m_w4g128_ref_on_mi355x.npz
timeout 600 python -m pytest tests/test_gpu_t3_fixture.py -q -m gpu > $O/t_fix.log 2>&1; echo "fixture rc=$?"; tail -30 $O/t_fix.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 600 $O/bench.err
python tools/kbench.py --n 7077888 > $O/kbench_opt_u1.json 2>&1; AR_INT_BWD_SMALL_U=2 python tools/kbench.py --n 7077888 > $O/kbench_opt_u2.json 2>&1
head -c 700 $O/kbench_opt_u1.json; echo; head -c 700 $O/kbench_opt_u2.json; echo
timeout 600 python bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 1 --warmup 1 --no-extras --no-cpu-baseline > $O/mixtral_mxfp4_fused.json 2> $O/mixtral_mxfp4_fused.err; echo "mix fused rc=$?"; tail -c 400 $O/mixtral_mxfp4_fused.err; head -c 1500 $O/mixtral_mxfp4_fused.json
timeout 600 python bench.py --workload mixtral-8x7b-hf --scheme MXFP4 --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-fused-block > $O/mixtral_mxfp4_module.json 2> $O/mixtral_mxfp4_module.err; echo "mix module rc=$?"; head -c 600 $O/mixtral_mxfp4_module.json
