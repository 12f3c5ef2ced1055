set -x
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r3a; mkdir -p $O
python -c "import torch; print(torch.cuda.get_device_name(0))"
timeout 1200 python -m pytest tests/test_gpu_fused_block.py -x -q -m gpu > $O/t_fused.log 2>&1; echo "fused rc=$?"
tail -15 $O/t_fused.log
timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "best_loss or graph or three_fused" > $O/t_kern.log 2>&1; echo "kern rc=$?"; tail -3 $O/t_kern.log
timeout 1500 python tests/t3_baseline_shapes.py --out $O/t3.json --fixture $O/t3_fixture.npz > $O/t3.log 2>&1; echo "t3 rc=$?"
tail -c 3000 $O/t3.log
cp $O/t3_fixture.npz tests/golden/t3_opt125m_w4g128_ref_on_mi355x.npz
timeout 600 python -m pytest tests/test_gpu_t3_fixture.py -q -m gpu > $O/t_fix.log 2>&1; echo "fixture rc=$?"; tail -30 $O/t_fix.log
timeout 900 python bench.py --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.err; cat $O/bench.json | head -c 6000
