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


def _weights(rng, G, gs, dt):
    """Per-group magnitudes log-uniform over nine decades plus the structural corner cases."""
    mag = 10.0 ** rng.uniform(-7, 2, size=(G, 1))
    w = rng.standard_normal((G, gs)) * mag
    kinds = rng.integers(0, 10, size=G)
    w[kinds == 0] = 0.0
    w[kinds == 1] = np.abs(w[kinds == 1])
    w[kinds == 2] = -np.abs(w[kinds == 2])
    tie = kinds == 3
    w[tie, 0] = -np.abs(w[tie]).max(axis=1)                       # |min| == |max|
    w[tie, 1] = np.abs(w[tie]).max(axis=1)
    return torch.from_numpy(w.astype(np.float32)).to(dt)


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
