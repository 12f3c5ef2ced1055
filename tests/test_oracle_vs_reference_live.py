"""Randomised live cross-check of the C oracle against the REFERENCE's own quant functions (build container only: the
GPU box has no /root/reference).  The committed goldens pin fixed configurations; here shapes, group sizes, bit widths,
dtypes and value ranges (1e-7 .. 1e2, zero groups, one-signed groups, ties) are drawn from seeds, and the forward results
must be bit-identical, the rounding-value gradient bit-identical, the range gradients sign-identical."""
import os
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from random_cases import group_values

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")

TD = {orc.DT_BF16: torch.bfloat16, orc.DT_F16: torch.float16, orc.DT_F32: torch.float32}


def _ref():
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)


def _bits(t):
    t = t.detach().contiguous()
    return t.view(torch.int16).numpy().view(np.uint16).copy() if t.dtype in (torch.bfloat16, torch.float16) else t.float().numpy().copy()


_weights = group_values


@pytest.mark.parametrize("seed", range(24))
def test_int_qdq_oracle_equals_the_reference_functions_on_random_cases(seed):
    _ref()
    from auto_round.data_type.int import quant_tensor_asym, quant_tensor_sym

    rng = np.random.default_rng(1000 + seed)
    bits = int(rng.choice([2, 3, 4, 8]))
    gs = int(rng.choice([16, 32, 64, 128, 40, 256]))
    sym = bool(rng.integers(0, 2))
    w_dt = [orc.DT_BF16, orc.DT_BF16, orc.DT_F16, orc.DT_F32][int(rng.integers(0, 4))]
    s_dt = orc.DT_F16 if rng.integers(0, 4) else orc.DT_F32
    G = int(rng.integers(8, 96))
    thresh = 1e-8 if s_dt == orc.DT_F32 else 1e-5
    W = _weights(rng, G, gs, TD[w_dt])
    if w_dt == orc.DT_F16:
        W = W.clamp(-6e4, 6e4)
    V = torch.from_numpy(((rng.random((G, gs)) - 0.5) * 1.4).astype(np.float32)).requires_grad_(True)
    ms = torch.from_numpy(rng.uniform(0.0, 1.0, G).astype(np.float32))
    Ms = torch.from_numpy(rng.uniform(0.0, 1.0, G).astype(np.float32))
    ms[:2], Ms[:2] = 1.0, 1.0
    ms[2], Ms[2] = 0.0, 0.0
    ms.requires_grad_(True), Ms.requires_grad_(True)
    wmin, wmax = torch.clamp(W.min(1)[0], max=0), torch.clamp(W.max(1)[0], min=0)
    fn = quant_tensor_sym if sym else quant_tensor_asym
    Wq, scale, zp = fn(W, bits=bits, group_size=gs, v=V, min_scale=ms, max_scale=Ms, scale_dtype=TD[s_dt], tensor_min=wmin,
                       tensor_max=wmax, q_scale_thresh=thresh)
    dWq = torch.from_numpy((rng.standard_normal((G, gs)) * 1e-3).astype(np.float32)).to(TD[w_dt])
    dWq.view(-1)[::53] = 0
    Wq.backward(dWq)

    Wb, Vn = _bits(W).reshape(-1), V.detach().numpy().reshape(-1)
    omin, omax = orc.group_minmax(Wb, w_dt, G, gs)
    assert np.array_equal(omin, _bits(wmin)) and np.array_equal(omax, _bits(wmax))
    a = (Wb, Vn, omin, omax, ms.detach().numpy(), Ms.detach().numpy(), G, gs, bits, int(sym), w_dt, s_dt, thresh)
    oWq, oscale, ozp = orc.qdq_int_fwd(*a)
    tag = f"bits={bits} gs={gs} sym={sym} w_dt={w_dt} s_dt={s_dt} G={G}"
    assert np.array_equal(oscale, _bits(scale.reshape(-1))), tag
    ref_zp = np.full(G, float(zp), np.float32) if not isinstance(zp, torch.Tensor) else zp.detach().reshape(-1).float().numpy()
    assert np.array_equal(ozp, ref_zp), tag
    assert np.array_equal(oWq, _bits(Wq).reshape(-1)), tag
    dV, dmin, dmax = orc.qdq_int_bwd(_bits(dWq).reshape(-1), *a)
    assert np.array_equal(dV.view(np.uint32), V.grad.numpy().reshape(-1).view(np.uint32)), tag
    for mine, ref in ((dmin, ms.grad.numpy()), (dmax, Ms.grad.numpy())):
        ok = np.isfinite(ref)
        big = np.abs(ref[ok]).max() if ok.any() else 0.0
        bad = ok & (np.sign(mine) != np.sign(ref))
        assert np.all(np.abs(ref[bad]) <= 1e-4 * big) and bad.mean() < 0.02, (tag, int(bad.sum()))


@pytest.mark.parametrize("seed", range(12))
def test_fp4_qdq_oracle_equals_the_reference_functions_on_random_cases(seed):
    _ref()
    from auto_round.data_type.mxfp import quant_mx
    from auto_round.data_type.nvfp import nv_fp4

    rng = np.random.default_rng(2000 + seed)
    nv = bool(seed % 2)
    gs = 16 if nv else 32
    w_dt = orc.DT_BF16 if seed % 3 else orc.DT_F16
    G = int(rng.integers(8, 64))
    W = _weights(rng, G, gs, TD[w_dt])
    if w_dt == orc.DT_F16:
        W = W.clamp(-6e4, 6e4)
    V = torch.from_numpy(((rng.random((G, gs)) - 0.5) * 1.4).astype(np.float32))
    Ms = torch.from_numpy(rng.uniform(0.3, 1.0, G).astype(np.float32))
    Wb = _bits(W).reshape(-1)
    if nv:
        from auto_round.data_type.nvfp import calculate_gparam

        gsc = orc.nvfp4_global_scale(Wb, w_dt)
        assert np.float32(gsc) == np.float32(calculate_gparam(W, gs).item())
        Wq, scale, _ = nv_fp4(W, bits=4, group_size=gs, v=V, max_scale=Ms, global_scale=torch.tensor(gsc))
        oWq, oscale = orc.qdq_nvfp4_fwd(Wb, V.numpy().reshape(-1), Ms.numpy(), gsc, G, gs, w_dt)[:2]
        assert np.array_equal(np.asarray(oscale, np.float32), scale.reshape(-1).float().numpy()), "nvfp4 group scales"
    else:
        Wq, exp, _ = quant_mx(W, bits=4, group_size=gs, v=V, max_scale=Ms, data_type="mx_fp")
        oWq = orc.qdq_mxfp4_fwd(Wb, V.numpy().reshape(-1), Ms.numpy(), G, gs, w_dt)[0]
    assert np.array_equal(oWq, _bits(Wq).reshape(-1)), f"nv={nv} G={G} w_dt={w_dt}"


@pytest.mark.parametrize("seed", range(12))
def test_packers_equal_the_reference_quantlinears_on_random_cases(seed):
    """qlinear_torch_zp / qlinear_torch `QuantLinear.pack` and the AWQ `WQLinear_GEMM.from_linear` on a random baked layer."""
    _ref()
    import auto_round_extension.torch.qlinear_torch as plain
    import auto_round_extension.torch.qlinear_torch_zp as zpmod
    from auto_round.data_type.int import quant_tensor_asym, quant_tensor_sym

    rng = np.random.default_rng(3000 + seed)
    bits = int(rng.choice([2, 3, 4, 8]))
    gs = int(rng.choice([32, 64, 128]))
    sym = bool(rng.integers(0, 2))
    out_f, in_f = 32 * int(rng.integers(1, 5)), gs * int(rng.integers(1, 4)) * (2 if gs == 32 else 1)
    if in_f % 32:
        in_f *= 2
    lin = torch.nn.Linear(in_f, out_f, bias=False)
    W = _weights(rng, out_f * in_f // gs, gs, torch.bfloat16).reshape(out_f, in_f)
    V = torch.from_numpy((rng.random((out_f * in_f // gs, gs)) - 0.5).astype(np.float32))
    Wq, scale, zp = (quant_tensor_sym if sym else quant_tensor_asym)(W, bits=bits, group_size=gs, v=V)
    lin.weight.data = Wq.detach().clone()
    scale2d = scale.reshape(out_f, -1)
    zp2d = zp.reshape(out_f, -1) if isinstance(zp, torch.Tensor) else zp
    ozp = zp2d.float().numpy() if isinstance(zp2d, torch.Tensor) else float(zp2d)
    Wb, sb = _bits(lin.weight.data).reshape(-1), _bits(scale2d).reshape(-1)
    for mod, off in ((zpmod, 1), (plain, 0)):
        ql = mod.QuantLinear(bits, gs, in_f, out_f, False)
        ql.device = "cpu"
        ql.pack(lin, scale2d.clone(), zp2d.clone() if isinstance(zp2d, torch.Tensor) else zp2d, None, device="cpu")
        qw, qz, st = orc.pack_int(Wb, sb, ozp, out_f, in_f, gs, bits, orc.DT_BF16, orc.DT_F16, zp_off=off)
        tag = f"bits={bits} gs={gs} sym={sym} {out_f}x{in_f} off={off}"
        assert np.array_equal(qw, ql.qweight.numpy()), tag
        assert np.array_equal(qz, ql.qzeros.numpy()), tag
        assert np.array_equal(st, _bits(ql.scales)), tag
    if bits == 4:
        from auto_round.export.export_to_awq.utils import WQLinear_GEMM

        z_awq = zp2d.t().contiguous().to(torch.float32) if isinstance(zp2d, torch.Tensor) else zp2d
        aq = WQLinear_GEMM.from_linear(lin, bits, gs, scales=scale2d.t().contiguous(), zeros=z_awq, device="cpu")
        qw, qz, st = orc.pack_awq(Wb, sb, ozp, out_f, in_f, gs)
        assert np.array_equal(qw, aq.qweight.numpy()) and np.array_equal(qz, aq.qzeros.numpy()) and np.array_equal(st, _bits(aq.scales))


@pytest.mark.parametrize("seed", range(16))
def test_int_activation_fake_quant_equals_the_reference_on_random_cases(seed):
    """quant_tensor_sym / quant_tensor_asym on activations as `WrapperLinear._qdq_act` calls them (v = 0, 0-dim scales)."""
    _ref()
    from auto_round.data_type.int import quant_tensor_asym, quant_tensor_sym

    rng = np.random.default_rng(4000 + seed)
    bits = int(rng.choice([4, 8]))
    hidden = int(rng.choice([128, 256, 384]))
    gs = int(rng.choice([32, 128, -1]))
    sym = bool(seed % 2)
    dt = orc.DT_F16 if seed % 4 == 3 else orc.DT_BF16
    g = hidden if gs == -1 else gs
    T = int(rng.integers(3, 24))
    x = _weights(rng, T * hidden // g, g, torch.float32).clamp(-3e4, 3e4).reshape(T, hidden).to(TD[dt]).requires_grad_(True)
    one = torch.tensor(1.0)
    xq, scale, zp = (quant_tensor_sym if sym else quant_tensor_asym)(
        x, bits=bits, group_size=gs, v=0, min_scale=one.clone(), max_scale=one.clone(), scale_dtype=torch.float16, tensor_max=None,
        q_scale_thresh=1e-5)
    dy = torch.from_numpy((rng.standard_normal((T, hidden)) * 1e-2).astype(np.float32)).to(TD[dt])
    xq.backward(dy)
    xb, G = _bits(x).reshape(-1), T * hidden // g
    tag = f"bits={bits} gs={gs} hidden={hidden} sym={sym} dt={dt} T={T}"
    if sym:
        oxq, os_ = orc.int_act_fwd(xb, G, g, bits, a_dt=dt)
        odx = orc.int_act_bwd(_bits(dy).reshape(-1), xb, G, g, bits, a_dt=dt)
    else:
        oxq, os_, ozp = orc.int_act_asym_fwd(xb, G, g, bits, a_dt=dt)
        assert np.array_equal(ozp, zp.detach().reshape(-1).float().numpy()), tag
        odx = orc.int_act_asym_bwd(_bits(dy).reshape(-1), xb, G, g, bits, a_dt=dt)
    assert np.array_equal(os_, _bits(scale.reshape(-1))), tag
    assert np.array_equal(oxq, _bits(xq).reshape(-1)), tag
    ref = _bits(x.grad).reshape(-1)
    same = odx == ref
    finite = np.isfinite(orc.from_bits(ref, TD[dt]).float().numpy())
    assert same[finite].mean() >= (1.0 if sym else 0.998), (tag, float(same[finite].mean()))


@pytest.mark.parametrize("seed", range(8))
def test_init_scale_searches_equal_the_reference_on_random_cases(seed):
    """search_scales (+ the search_int threshold clamp) and the fp4 coefficient searches of the algorithm extension, with a
    random importance matrix, on weights of ordinary magnitude (the searches compare fp32 loss sums, so near-ties between
    two candidates may resolve differently; everything else must be identical)."""
    _ref()
    from auto_round.data_type.int import search_scales
    from auto_round.data_type.mxfp import search_mx_scale
    from auto_round.data_type.nvfp import search_nvfp4_scale
    from auto_round.data_type.utils import reshape_imatrix_for_weight, search_optimized_init_scale

    rng = np.random.default_rng(5000 + seed)
    cols = int(rng.choice([128, 256, 384]))
    rows = int(rng.integers(4, 24))
    imatrix = torch.from_numpy(((rng.random(cols) * 4.0 + 0.01) ** 2 * 100.0).astype(np.float32))
    imatrix[int(rng.integers(0, cols))] = 0.0
    W = torch.from_numpy((rng.standard_normal((rows, cols)) * 10.0 ** rng.uniform(-3, 0)).astype(np.float32)).to(torch.bfloat16)
    W[0, :32] = 0.0
    Wb = _bits(W).reshape(-1)
    if seed % 2:
        nb, gs = [(4, 128), (2, 32), (3, 64), (8, 128)][seed // 2 % 4]
        Wg = W.reshape(-1, gs)
        qw = reshape_imatrix_for_weight(imatrix, Wg, gs)
        raw_ref = _bits(search_scales(Wg, nb, qw).reshape(-1))
        init_ref = _bits(search_optimized_init_scale(Wg, "int_sym", nb, qw, 1e-5).reshape(-1))
        ones_ref = _bits(search_optimized_init_scale(Wg, "int_sym", nb, None, 1e-5).reshape(-1))
        raw, init = orc.search_int_scale(Wb, Wg.shape[0], gs, nb, qw_row=imatrix.numpy(), groups_per_row=cols // gs)
        _, ones = orc.search_int_scale(Wb, Wg.shape[0], gs, nb)
        assert (raw == raw_ref).mean() >= 0.99 and (init == init_ref).mean() >= 0.99 and (ones == ones_ref).mean() >= 0.99, (nb, gs)
    else:
        nv = bool(seed // 2 % 2)
        gs, mode = (16, 1) if nv else (32, 0)
        Wg = W.reshape(-1, gs)
        qw = reshape_imatrix_for_weight(imatrix, Wg, gs)
        fn = search_nvfp4_scale if nv else search_mx_scale
        gsc = orc.nvfp4_global_scale(Wb) if nv else 1.0
        for q, row in ((qw, imatrix.numpy()), (torch.ones_like(Wg, dtype=torch.float32), None)):
            ref = fn(Wg, 4, q).reshape(-1).float().numpy()
            best = orc.search_fp4_scale(Wb, Wg.shape[0], gs, mode, qw_row=row, groups_per_row=cols // gs, global_scale=gsc)
            assert (best == ref).mean() >= 0.99, (nv, float((best == ref).mean()))


@pytest.mark.parametrize("seed", range(8))
def test_losses_equal_the_reference_on_random_cases(seed):
    """SignRoundQuantizer._get_loss (plain / valid-token mask) and SignRoundV2Quantizer._get_loss (outlier-suppressed) with
    their x1000 backward, on random bf16 / fp16 activations."""
    _ref()
    from types import SimpleNamespace

    from auto_round.algorithms.quantization.sign_round.quantizer import SignRoundQuantizer as RefQ
    from auto_round.algorithms.quantization.sign_roundv2.quantizer import SignRoundV2Quantizer as V2

    rng = np.random.default_rng(6000 + seed)
    dt = orc.DT_F16 if seed % 3 == 2 else orc.DT_BF16
    B, S, H = int(rng.integers(1, 5)), int(rng.choice([16, 48, 64])), int(rng.choice([64, 128]))
    pred = torch.from_numpy(rng.standard_normal((B, S, H)).astype(np.float32) * 10.0 ** rng.uniform(-2, 1)).to(TD[dt])
    ref = (pred.float() + 0.05 * pred.float().abs().mean() * torch.from_numpy(rng.standard_normal((B, S, H)).astype(np.float32))).to(TD[dt])
    mask = [torch.ones(1, S, dtype=torch.long) for _ in range(B)]
    for m in mask:
        m[0, -1] = 0
        m[0, int(rng.integers(0, S - 1))] = 0
    mflat = torch.cat(mask).reshape(-1).numpy().astype(np.uint8)
    fake = SimpleNamespace(model_context=SimpleNamespace(amp=True, amp_dtype=TD[dt]))
    for m, mo in ((None, None), (mask, mflat)):
        p2 = pred.detach().clone().requires_grad_(True)
        loss = RefQ._get_loss(fake, p2, ref, list(range(B)), torch.nn.MSELoss(), "cpu", m)
        (loss * 1000).backward()
        lo, d = orc.mse_fwd_bwd(_bits(pred).reshape(-1), _bits(ref).reshape(-1), act_dt=dt, token_mask=mo, row_len=H)
        assert abs(lo - loss.item()) <= 2e-6 * abs(loss.item())
        a, b = orc.from_bits(d, TD[dt]).float().numpy(), p2.grad.float().numpy().reshape(-1)
        assert np.array_equal(a, b), (seed, m is None)
    if B * S * H >= 4000:        # the reference's top-0.1% drop needs at least a handful of elements to drop
        fake2 = SimpleNamespace(_use_outlier_suppressed_loss=True, amp=True, amp_dtype=TD[dt])
        p2 = pred.detach().clone().requires_grad_(True)
        loss = V2._get_loss(fake2, p2, ref, list(range(B)), torch.nn.MSELoss(), "cpu", None)
        (loss * 1000).backward()
        lo, d, dropped = orc.outlier_mse_fwd_bwd(_bits(pred).reshape(-1), _bits(ref).reshape(-1), act_dt=dt)
        assert dropped == max(1, B * S * H // 1000)
        assert abs(lo - loss.item()) <= 2e-3 * abs(loss.item())
        a, b = orc.from_bits(d, TD[dt]).float().numpy(), p2.grad.float().numpy().reshape(-1)
        assert (a != b).sum() <= 4 and (a == 0).sum() == (b == 0).sum()
