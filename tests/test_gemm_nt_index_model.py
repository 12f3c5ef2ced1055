"""CPU: the index arithmetic of csrc/ar_gemm_nt.hip (LDS-DMA source swizzle, lane-linear LDS image, fragment read addresses, epilogue
ownership) restated in tools/gemm_nt_index_model.py -- every lane gets the operand rows / k range its MFMA wants, no bank conflicts in
any 16-lane group of a ds_read_b128, every output element stored exactly once.  The bits themselves are checked on the GPU
(tools/gpu/r05_gemm_nt_probe.py, tests/test_gpu_gemm_nt.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_dma_swizzle_fragment_reads_and_epilogue_are_consistent():
    import gemm_nt_index_model as m

    worst, covered = m.main()
    assert worst == 1 and covered


def test_the_model_restates_the_kernels_formulas():
    """the constants the model uses appear in the kernel source (a change of one side only must be noticed)"""
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "auto_round_amd", "csrc", "ar_gemm_nt.hip")).read()
    for frag in ("pc ^ ((4 * j + (lane >> 4)) & 7)", "l31 * 128 + 16 * ((4 * ah + 2 * u + h) ^ s)", "(l31 >> 1) & 7",
                 "lds0 + 65536 + (wc >> 1) * 32768 + (wc & 1) * 8192 + X", "lds0 + wr * 32768 + wc * 32 * 128",
                 # the 16x16x32 form
                 "l15 * 128 + 16 * ((4 * ah + kg) ^ s)", "adB16[ah] = lds0 + 65536 + (wc >> 1) * 32768 + (wc & 1) * 8192 + X;",
                 # nt2
                 "const int row = wave * 64 + 8 * j + (lane >> 3);", "lds0 + (wave >> 1) * 32768 + (wave & 1) * 64 * 128", "l31 * 128 + 16 * ((2 * u + h) ^ s)",
                 "adB[u] = lds0 + 65536 + wc * 32768 + X;"):
        assert frag in src, frag
