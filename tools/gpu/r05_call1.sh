#!/bin/bash
# round 5, GPU call 1: search kernels vs torch on the GPU, bare MFMA peak (zeros / random), reference-made digests + two-run fixtures
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp PYTHONDONTWRITEBYTECODE=1
O=gpurun_out/r05; mkdir -p $O/golden
./tools/bin/mfma_peak 20000 > $O/mfma_peak.jsonl 2>&1; tail -3 $O/mfma_peak.jsonl
timeout 600 python -m pytest tests/test_gpu_autograd_bits.py -q -x -k "search" > $O/search_bits.log 2>&1; tail -5 $O/search_bits.log
timeout 900 python tests/t3_baseline_shapes.py --cases llama8b_w2g32_sym_algext_200,llama8b_mxfp4_algext_200,llama8b_nvfp4_algext_200 \
    --digest-dir $O/golden --skip-alone --variants module,exact --out $O/t3_algext_sym.json > $O/t3_algext_sym.log 2>&1
echo "t3 algext rc=$?"; grep -v "amdgpu.ids\|layer_idx" $O/t3_algext_sym.log | tail -5 | cut -c1-1500
timeout 600 python tests/t3_baseline_shapes.py --cases opt125m_w4g128 --ref-twice opt125m_w4g128 --stat-fixture-dir $O/golden \
    --skip-alone --variants module,fused --out $O/t3_opt_twice.json > $O/t3_opt_twice.log 2>&1
echo "t3 opt rc=$?"; grep -v "amdgpu.ids\|layer_idx" $O/t3_opt_twice.log | tail -3 | cut -c1-1500
timeout 900 python tests/t3_baseline_shapes.py --cases mixtral8x7b_mxfp4_100,mixtral8x7b_nvfp4_100 --ref-twice mixtral8x7b_mxfp4_100,mixtral8x7b_nvfp4_100 \
    --stat-fixture-dir $O/golden --skip-alone --variants module,fused --out $O/t3_mixtral_twice.json > $O/t3_mixtral_twice.log 2>&1
echo "t3 mixtral rc=$?"; grep -v "amdgpu.ids\|layer_idx" $O/t3_mixtral_twice.log | tail -3 | cut -c1-1500
ls -la $O/golden
