cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/aotp
rocprofv3 --kernel-trace --output-format csv -d /tmp/aotp -- python $GRAFT_REPO_ROOT/tools/gpu/r06_aotriton_configs.py 2>&1 | grep -v amdgpu.ids | tail -5
f=$(find /tmp/aotp -name '*kernel_trace.csv' | head -1); echo $f
python $GRAFT_REPO_ROOT/tools/gpu/r06_trace_kernels.py $f attn_fwd,bwd_kernel,bwd_pre,bwd_post | tee $GRAFT_REPO_ROOT/gpurun_out/r06/aotriton_configs.txt
