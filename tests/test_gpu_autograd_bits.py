"""Bit-level parity of the fake-quant backward kernels with torch autograd ON THE GPU (round 3).

The reference's quant functions are plain torch code, so "what the reference computes" for the three gradients of a wrapped layer --
d loss / d V, d min_scale, d max_scale -- is whatever torch's autograd produces for them on the device the reference runs on: its
elementwise ops in fp32, its reduction kernel's summation order for the per-group sums (`sum_to_size`), its `tensor / python_scalar`
evaluated as a multiplication by the reciprocal.  `oracle/torch_ref.py` restates those functions and is pinned to the reference on
the CPU (tests/test_torch_ref.py, live against /root/reference); here the same restatement runs on the MI355X under autograd and
the HIP kernels must reproduce every gradient BIT FOR BIT -- which is what lets 200 sign-SGD iterations of a Llama-3-8B block come
out identical to the reference's (tests/test_gpu_t3_fixture.py, profiles/r03_t3_baseline_shapes.json).

Before round 3 only dV was bit-exact; the scale gradients were "sign-exact" (another summation order, a true division by maxq, a
closed-form MX element derivative)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _bits_equal(a, b):
    a, b = a.float().contiguous().view(torch.int32), b.float().contiguous().view(torch.int32)
    return float((a == b).float().mean())


def _values_equal(a, b):          # bit-equal, signed zeros apart
    return bool(torch.equal(a.float(), b.float()))


def _problem(out_f, in_f, gs, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = (torch.randn(out_f, in_f, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    G = out_f * in_f // gs
    V = (torch.rand(G, gs, generator=g, device="cuda") - 0.5).requires_grad_(True)
    mn = (0.85 + 0.15 * torch.rand(G, generator=g, device="cuda")).requires_grad_(True)
    mx = (0.85 + 0.15 * torch.rand(G, generator=g, device="cuda")).requires_grad_(True)
    dWq = (torch.randn(out_f, in_f, generator=g, device="cuda") * 1e-3).to(torch.bfloat16)
    return W, G, V, mn, mx, dWq


@pytest.mark.parametrize("bits,gs,sym", [(4, 128, True), (4, 32, True), (2, 32, False), (4, 128, False), (3, 64, True), (8, 128, True),
                                         (2, 64, False), (4, 256, True)])
def test_int_backward_is_bit_identical_to_torch_autograd_on_the_gpu(bits, gs, sym):
    """W2 / W3 / W4 / W8, symmetric and asymmetric, the group sizes of the BASELINE configurations (128, 32) and their neighbours."""
    from auto_round_amd import ops
    from oracle import torch_ref as tr

    W, G, V, mn, mx, dWq = _problem(2048, 4096, gs)
    Wg = W.reshape(-1, gs)
    wmin, wmax = torch.clamp(Wg.min(1)[0], max=0), torch.clamp(Wg.max(1)[0], min=0)
    Wq, s, zp = tr.qdq_int(W, bits, gs, sym, V, mn, mx, wmin, wmax)
    Wq.backward(dWq)
    Wq_k, s_k, zp_k = ops.qdq_int_fwd(W.view(-1), V.detach().reshape(-1).contiguous(), wmin, wmax, mn.detach(), mx.detach(), gs=gs, bits=bits,
                                      sym=int(sym), want_scale=True)
    assert torch.equal(Wq_k.view(torch.int16), Wq.detach().reshape(-1).view(torch.int16))
    assert torch.equal(s_k.view(torch.int16), s.detach().reshape(-1).view(torch.int16))
    dV, dmin, dmax = ops.qdq_int_bwd(dWq.view(-1), W.view(-1), V.detach().reshape(-1).contiguous(), wmin, wmax, mn.detach(), mx.detach(),
                                     gs=gs, bits=bits, sym=int(sym))
    assert _values_equal(dV, V.grad.reshape(-1)), _bits_equal(dV, V.grad.reshape(-1))
    if gs in (32, 64, 128):        # the row lengths whose reduction order is mirrored exactly (torch's kernel: <= 64 one element per
        assert _values_equal(dmin, mn.grad), (bits, gs, sym, _bits_equal(dmin, mn.grad))      # thread, 128 one float4 per thread)
        assert _values_equal(dmax, mx.grad), (bits, gs, sym, _bits_equal(dmax, mx.grad))
    else:                          # longer rows: several vectors per thread -- same value to rounding, sign-exact
        for mine, ref in ((dmin, mn.grad), (dmax, mx.grad)):
            assert _bits_equal(mine, ref) > 0.9
            bad = torch.sign(mine) != torch.sign(ref)
            assert float(bad.float().mean()) < 1e-3


@pytest.mark.parametrize("kind", ["mxfp4", "nvfp4"])
def test_fp4_backward_is_bit_identical_to_torch_autograd_on_the_gpu(kind):
    from auto_round_amd import ops
    from oracle import torch_ref as tr

    gs = 32 if kind == "mxfp4" else 16
    W, G, V, _, mx, dWq = _problem(2048, 4096, gs, seed=1)
    if kind == "mxfp4":
        Wq, _ = tr.qdq_mxfp4(W, gs, V, mx)
        gsc = None
    else:
        gsc = tr.nvfp4_global_scale(W)
        Wq, _ = tr.qdq_nvfp4(W, gs, V, mx, gsc)
    Wq.backward(dWq)
    absmax, _ = ops.group_absmax(W.view(-1), gs)
    gs_dev = None if gsc is None else torch.as_tensor(gsc, dtype=torch.float32, device="cuda").reshape(1)
    dV, dmax = ops.qdq_fp4_bwd_sgd_(dWq.view(-1), W.view(-1), V.detach().reshape(-1).contiguous(), absmax, mx.detach(), mode=0 if kind == "mxfp4" else 1,
                                    gs=gs, global_scale=gs_dev, want_grads=True)
    assert _values_equal(dV, V.grad.reshape(-1)), _bits_equal(dV, V.grad.reshape(-1))
    assert _values_equal(dmax, mx.grad), _bits_equal(dmax, mx.grad)


# ---- round 5: the init-scale searches of the algorithm extension, against the torch restatement ON THE GPU -------------------------
def _search_problem(out_f, in_f, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    W = (torch.randn(out_f, in_f, generator=g, device="cuda") * 0.02).to(torch.bfloat16)
    # an importance row the way the imatrix hooks leave it: sums of squares over many tokens, a few decades of dynamic range
    im = (torch.randn(in_f, generator=g, device="cuda") ** 2 * torch.exp(2.0 * torch.randn(in_f, generator=g, device="cuda")) * 4096.0).float()
    return W, im


@pytest.mark.parametrize("bits,gs", [(2, 32), (4, 128), (4, 32), (3, 64), (8, 128)])
@pytest.mark.parametrize("weighted", [True, False], ids=["imatrix", "ones"])
def test_int_init_scale_search_is_bit_identical_to_torch_on_the_gpu(bits, gs, weighted):
    """search_scales (data_type/int.py:24-86) is a chain of `loss < best_loss` over ~180-400 candidates per group: one low bit of a
    group's loss decides a near-tie, and the searched scale seeds the whole trajectory of an algorithm-extension run (W2G32 sym, the
    recipe the reference recommends).  The kernel adds a group's errors in the association torch's reduction kernel uses for
    `sum(dim=-1)` of an [G, gs] fp32 tensor on this GPU -- every group must come out with the same scale, bit for bit."""
    from auto_round_amd import ops
    from oracle import torch_ref as tr

    out_f, in_f = 1024, 4096
    W, im = _search_problem(out_f, in_f, seed=bits * 1000 + gs)
    Wg = W.reshape(-1, gs)
    qw = im.view(1, -1).expand(out_f, -1).reshape(-1, gs) if weighted else None
    want = tr.search_int_scale(Wg, bits, qw=qw).reshape(-1)
    got = ops.search_int_scale(W.view(-1), gs=gs, bits=bits, qw_row=im if weighted else None, groups_per_row=in_f // gs)
    same = got.view(torch.int16) == want.view(torch.int16)
    assert bool(same.all()), (bits, gs, weighted, float(same.float().mean()))


@pytest.mark.parametrize("kind", ["mxfp4", "nvfp4"])
@pytest.mark.parametrize("weighted", [True, False], ids=["imatrix", "ones"])
def test_fp4_init_scale_search_is_bit_identical_to_torch_on_the_gpu(kind, weighted):
    """search_mx_scale (mxfp.py:103-170: coefficients 1, 0.5, 2) and search_nvfp4_scale (nvfp.py:329-386: 1.0, then 0.50 ... 1.51)."""
    from auto_round_amd import ops
    from oracle import torch_ref as tr

    out_f, in_f = 1024, 4096
    gs, mode = (32, 0) if kind == "mxfp4" else (16, 1)
    W, im = _search_problem(out_f, in_f, seed=7 + mode)
    Wg = W.reshape(-1, gs)
    qw = im.view(1, -1).expand(out_f, -1).reshape(-1, gs) if weighted else None
    want = (tr.search_mx_coeff(Wg, qw=qw) if mode == 0 else tr.search_nv_coeff(Wg, qw=qw)).reshape(-1)
    absmax, tmax = ops.group_absmax(W.view(-1), gs, want_tensor_max=True)
    gsc = (448.0 * 6.0 * (1.0 / tmax)).to(torch.float32) if mode else None
    cand = torch.tensor(ops.fp4_search_candidates(mode), dtype=torch.float32, device="cuda")
    got = ops.search_fp4_scale(W.view(-1), absmax, cand, mode=mode, gs=gs, qw_row=im if weighted else None, groups_per_row=in_f // gs,
                               global_scale=gsc)
    same = got == want
    assert bool(same.all()), (kind, weighted, float(same.float().mean()))
