"""Launch the weight-gradient GEMM variants a few times on one shape (for rocprofv3 --pmc passes; see gpurun_variants/*.sh)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from auto_round_amd import ops  # noqa: E402
from auto_round_amd._lib import load  # noqa: E402

M, N, K = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (14336, 4096, 16384)))
dev = torch.device("cuda")
torch.manual_seed(0)
dY = torch.randn(K, M, device=dev).to(torch.bfloat16)
X = torch.randn(K, N, device=dev).to(torch.bfloat16)
out = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
lib = load()
for _ in range(3):
    torch.mm(dY.t(), X, out=out)
    for code in (11, 17):
        lib.ar_gemm_dw_config(code, 2)
        ops.gemm_dw(dY, X, out)
torch.cuda.synchronize()
