"""CPU: the index arithmetic of the weight-gradient GEMM kernels (csrc/ar_gemm.hip: k_gemm_dw4 on v_mfma_f32_32x32x16_bf16, k_gemm_dw6
on 16x16x32) restated in tools/gemm_dw_index_model.py -- LDS-DMA source swizzle, lane-linear LDS image, the hardware rule of the
transposing read, fragment addresses, epilogue ownership: every lane gets the columns / k range its MFMA operand wants, no bank
conflicts in either half-wave of a read, every output element stored exactly once.  The bits are checked on the GPU
(tests/test_gpu_fused_block.py, tools/gpu/r05_gemm_dw_m16_probe.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))


def test_both_mfma_shapes_read_the_operands_their_layout_wants_without_bank_conflicts():
    import gemm_dw_index_model as m

    out = m.main()
    assert out == {"32x32x16": (1, True), "16x16x32": (1, True)}
    assert m.worst_16x16x32_on_the_old_swizzle() == 2        # the new read pattern on the old swizzle: two-way conflicts


def test_the_model_restates_the_kernels_formulas():
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "auto_round_amd", "csrc", "ar_gemm.hip")).read()
    for frag in ("const int lchunk = (lane & 31) ^ (((drow & 3) << 2) | (((drow >> 3) & 1) << 1));",      # k_gemm_dw6: DMA source swizzle
                 "const int swz = (rowsel << 2) | ((q & 1) << 1);",
                 "const int rowoff = (q >> 1) * UNIT + (8 * (q & 1) + rowsel) * ROWB + (piece & 1) * 8;",
                 "const int chunk = wn * 16 + ni * 2 + (piece >> 1);", "const int chunk = wm * 8 + mi * 2 + (piece >> 1);",
                 "const int mrow = lane & 15, ncol = 4 * (lane >> 4);",
                 "const int lchunk = (lane & 31) ^ ((drow & 3) << 2);",                                      # k_gemm_dw4
                 "const int rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8;",
                 "const int chunk = wn * 16 + ni * 4 + (q & 1) * 2 + (piece >> 1);", "const int chunk = wm * 8 + mi * 4 + (q & 1) * 2 + (piece >> 1);"):
        assert frag in src, frag
