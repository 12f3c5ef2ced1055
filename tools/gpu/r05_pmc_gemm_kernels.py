"""A few launches of the first-party GEMM kernels on both MFMA shapes (weight gradient: k_gemm_dw4 / k_gemm_dw6; NT: variants 0 / 3) and of
the library kernels next to them, for one `rocprofv3 --pmc` pass (MFMA busy, wait breakdown, LDS bank conflicts):

    bash tools/gpu/run.sh pmc gemm_m16 "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" -- tools/gpu/r05_pmc_gemm_kernels.py
    python tools/gpu/r05_pmc_gemm_kernels.py --summarise gpurun_out/<tag>/gemm_m16_counter_collection.csv > profiles/r05_pmc_gemm_kernels.json
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        import r04_pmc_kernels as r4

        print(json.dumps(r4.summarise(sys.argv[2], names=("k_gemm_dw4", "k_gemm_dw6", "k_gemm_nt", "Cijk")), indent=1))
        return
    import torch

    from auto_round_amd import ops
    from auto_round_amd._lib import load

    lib = load()
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    bf = torch.bfloat16
    T = 16384
    for (M, N) in ((14336, 4096), (4096, 4096)):
        dY = (0.01 * torch.randn(T, M, device=dev, generator=g)).to(bf)
        X = torch.randn(T, N, device=dev, generator=g).to(bf)
        out = torch.empty(M, N, dtype=bf, device=dev)
        for code in (30, 32):
            lib.ar_gemm_dw_config(code, -1)
            for _ in range(3):
                ops.gemm_dw(dY, X, out, split=False)
        for _ in range(3):
            torch.mm(dY.t(), X, out=out)
        del dY, X, out
    lib.ar_gemm_dw_config(32, -1)
    for (M, N, K) in ((T, 4096, 4096), (T, 4096, 14336)):
        A = torch.randn(M, K, device=dev, generator=g).to(bf)
        B = (0.05 * torch.randn(N, K, device=dev, generator=g)).to(bf)
        out = torch.empty(M, N, dtype=bf, device=dev)
        for v in (0, 3):
            lib.ar_gemm_nt_config(v)
            for _ in range(3):
                ops.gemm_nt(A, B, out)
        for _ in range(3):
            torch.mm(A, B.t(), out=out)
        del A, B, out
    lib.ar_gemm_nt_config(3)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
