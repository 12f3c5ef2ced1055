"""The NT forward / input-gradient GEMM (csrc/ar_gemm_nt.hip, SURVEY 8 row f1 forward side) and the grouped expert GEMMs, through the
C ABI on the MI355X: against an fp32 product (the plain-PyTorch reference of the op), against the library GEMM behind `F.linear`
(hipBLASLt) bit for bit at the forward shapes of BASELINE configs[1] -- both sum K in ascending steps of 16 through
v_mfma_f32_32x32x16_bf16, which is what lets the kernel stand in for the library on the bit-identical path --, ragged row counts, and
run-to-run identical bits (the kernel keeps LDS-DMA in flight across barriers: a race shows as a flake)."""
import warnings

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rnd(shape, seed, scale=1.0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    return (torch.randn(shape, generator=g, device="cuda") * scale).to(torch.bfloat16)


def _bits(a, b):
    return int((a.contiguous().view(torch.int16) != b.contiguous().view(torch.int16)).sum())


@pytest.mark.parametrize("variant", [0, 1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (1000, 512, 384), (37, 256, 256), (4096, 1024, 4096), (16384, 4096, 4096)])
def test_gemm_nt_matches_fp32_and_the_library_and_is_reproducible(M, N, K, variant):
    from auto_round_amd import _lib, ops

    lib = _lib.load()
    keep = lib.ar_gemm_nt_config(-1)
    lib.ar_gemm_nt_config(variant)
    try:
        A, B = _rnd((M, K), 1), _rnd((N, K), 2, 0.05)          # asymmetric operands: a transposed tile or operand cannot pass
        out = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        assert ops.gemm_nt(A, B, out)
        rows = min(M, 1024)
        ref = A[:rows].float() @ B.float().t()
        err = (out[:rows].float() - ref).abs().max().item()
        lib_out = torch.mm(A, B.t())
        lib_err = (lib_out[:rows].float() - ref).abs().max().item()
        assert not torch.isnan(out.float()).any()
        assert err <= 1.5 * lib_err + 1e-3, (err, lib_err)          # one bf16 rounding of an fp32 sum, like the library's
        d = _bits(out, lib_out)
        if (M, N, K) == (16384, 4096, 4096):
            assert d == 0, f"{d} values differ from the library's forward GEMM at Llama-3-8B's o / q projection shape"
        elif d:
            warnings.warn(f"gemm_nt {M}x{N}x{K}: {d} values differ from the library's kernel for this shape (another summation split)")
        for _ in range(4):
            o2 = torch.empty_like(out)
            ops.gemm_nt(A, B, o2)
            assert _bits(o2, out) == 0, "two launches on the same operands returned different bits"
    finally:
        lib.ar_gemm_nt_config(keep)


def test_gemm_nt_takes_strided_operands_and_refuses_what_it_cannot_do():
    from auto_round_amd import ops

    big_a, big_b = _rnd((512, 1024), 3), _rnd((512, 1024), 4, 0.05)
    A, B = big_a[:, 256:640], big_b[:256, 512:896]                # column slices: unit inner stride, leading dimension 1024
    out_big = torch.zeros((512, 512), dtype=torch.bfloat16, device="cuda")
    out = out_big[:, 128:384]
    assert ops.gemm_nt(A, B, out)
    ref = A.float() @ B.float().t()
    assert (out.float() - ref).abs().max().item() <= 2e-2 * ref.abs().max().item()
    assert float(out_big[:, :128].abs().max()) == 0.0 and float(out_big[:, 384:].abs().max()) == 0.0      # nothing outside the slice
    assert ops.gemm_nt(_rnd((64, 128), 5), _rnd((100, 128), 6), torch.empty((64, 100), dtype=torch.bfloat16, device="cuda")) is False     # N % 256
    assert ops.gemm_nt(_rnd((64, 192), 5), _rnd((256, 192), 6), torch.empty((64, 256), dtype=torch.bfloat16, device="cuda")) is False     # K % 128
    assert ops.gemm_nt(A.float(), B.float(), out.float()) is False


@pytest.mark.parametrize("counts,N,K", [([300, 0, 1, 255, 257, 512, 100, 700], 512, 256), ([0, 0, 5], 256, 128), ([1100, 900, 1300, 796], 1024, 512)])
def test_grouped_gemms_equal_the_dense_kernels_group_by_group(counts, N, K):
    """ops.gemm_nt_grouped / ops.gemm_dw_grouped with device-side row offsets against one dense call per group: identical bits (same
    kernel, same summation order), empty groups, row counts that are multiples of nothing; a group without rows gets a zero gradient."""
    from auto_round_amd import ops

    E, R = len(counts), sum(counts)
    A = _rnd((R, K), 7)
    W = _rnd((E * N, K), 8, 0.05)
    row_off = torch.tensor([0] + torch.tensor(counts).cumsum(0).tolist(), dtype=torch.int32, device="cuda")
    b_off = torch.arange(E, dtype=torch.int64, device="cuda") * (N * K)
    out = torch.full((R, N), float("nan"), dtype=torch.bfloat16, device="cuda")
    assert ops.gemm_nt_grouped(A, W, out, row_off, b_off, N, K)
    assert not torch.isnan(out.float()).any()
    s = 0
    for e, c in enumerate(counts):
        if c:
            mine = torch.empty((c, N), dtype=torch.bfloat16, device="cuda")
            assert ops.gemm_nt(A[s:s + c], W[e * N:(e + 1) * N], mine)
            assert _bits(out[s:s + c], mine) == 0, e
            ref = A[s:s + c].float() @ W[e * N:(e + 1) * N].float().t()
            assert (out[s:s + c].float() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1e-3)
        s += c
    o2 = torch.empty_like(out)
    ops.gemm_nt_grouped(A, W, o2, row_off, b_off, N, K)
    assert _bits(o2, out) == 0
    # weight gradients: dW_e [N, K] = dY[rows_e]^T A[rows_e]
    dY = _rnd((R, N), 9, 0.01)
    dW = torch.full((E * N, K), float("nan"), dtype=torch.bfloat16, device="cuda")
    if K % 256:
        assert ops.gemm_dw_grouped(dY, A, dW, row_off, b_off, K) is False          # the gradient's width must be a multiple of 256 too
        return
    assert ops.gemm_dw_grouped(dY, A, dW, row_off, b_off, K)
    assert not torch.isnan(dW.float()).any()
    s = 0
    for e, c in enumerate(counts):
        got = dW[e * N:(e + 1) * N]
        if c == 0:
            assert float(got.abs().max()) == 0.0
        else:
            ref = dY[s:s + c].float().t() @ A[s:s + c].float()
            assert (got.float() - ref).abs().max().item() <= 2e-2 * max(ref.abs().max().item(), 1e-4)
            if c >= 96:
                mine = torch.empty((N, K), dtype=torch.bfloat16, device="cuda")
                if ops.gemm_dw(dY[s:s + c], A[s:s + c], mine, split=False):
                    assert _bits(got, mine) == 0, e
        s += c
