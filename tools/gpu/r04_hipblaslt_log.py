"""Which hipBLASLt solution serves the weight-gradient GEMM torch.mm(dY.t(), X) at the Llama-3-8B shapes?  Run with
HIPBLASLT_LOG_MASK=96 HIPBLASLT_LOG_FILE=<file> (bench + profile records: solution index, kernel name, split / stream-K arguments)."""
import sys

import torch

T = 16384
for name, (o, i) in dict(q=(4096, 4096), g=(14336, 4096), d=(4096, 14336), gu=(28672, 4096)).items():
    dY = torch.randn(T, o, device="cuda").to(torch.bfloat16)
    X = torch.randn(T, i, device="cuda").to(torch.bfloat16)
    print("== shape", name, o, i, flush=True)
    sys.stderr.write(f"== shape {name} {o}x{i}\n")
    torch.mm(dY.t(), X)
    torch.cuda.synchronize()
