"""floor(log2(x)) -- the MXFP4 shared exponent (auto_round/data_type/mxfp.py:90-91) -- on the inputs where two libms can
disagree: every float within +-8 ulp of a power of two.  Compares torch on the CPU (what the CPU goldens / the C oracle's glibc
log2f stand for) with torch on the GPU (what the reference computes when IT runs on the MI355X) and with this repository's fp4
kernel (its shared exponent read back from the scale output of ar_qdq_fp4_fwd on a one-group tensor per input)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_round_amd import ops  # noqa: E402

xs = []
for k in range(-120, 121):
    b = np.float32(2.0 ** k).view(np.uint32)
    for d in range(-8, 9):
        xs.append(np.uint32(int(b) + d).view(np.float32))
x = torch.tensor(np.array(xs, dtype=np.float32))
cpu = torch.floor(torch.log2(x))
gpu = torch.floor(torch.log2(x.cuda())).cpu()
# the kernel: one MXFP4 group of 32 per input whose absmax is x (fp32 tensor), max_scale = 1 -> scale_out = floor(log2 x) - 2
G = x.numel()
X = torch.zeros(G, 32, dtype=torch.float32, device="cuda")
X[:, 0] = x.cuda()
absmax, _ = ops.group_absmax(X.view(-1), 32)
ones = torch.ones(G, dtype=torch.float32, device="cuda")
_, scale = ops.qdq_fp4_fwd(X.view(-1), torch.zeros(G * 32, dtype=torch.float32, device="cuda"), absmax, ones, mode=0, gs=32, want_scale=True)
kern = scale.float().cpu() + 2.0        # for MXFP4 the scale output IS the shared exponent e = floor(log2 x) - 2 (csrc/ar_fp4.hip)
print(json.dumps({"check": "floor(log2(x)) for x within +-8 ulp of 2^k, k in [-120, 120]", "inputs": int(G),
                  "torch_gpu_vs_torch_cpu_mismatch": int((gpu != cpu).sum()), "kernel_vs_torch_gpu_mismatch": int((kern != gpu).sum()),
                  "kernel_vs_torch_cpu_mismatch": int((kern != cpu).sum()),
                  "note": "the reference on the GPU and the kernel use the same device libm; both differ from the CPU libm only for "
                          "inputs a few ulp BELOW a power of two, where the correctly rounded log2 is the integer itself"}))
