"""HIP kernels against the C oracle on the seeded wide-range inputs of tests/random_cases.py (the same draws the live test
holds the oracle to the reference on): per-group magnitudes from 1e-7 to 1e2, zero / one-signed / tied groups, every bit
width, odd group sizes, both scale dtypes.  Same bars as test_gpu_kernels.py: bit-exact forward, dV, packed words."""
import numpy as np
import pytest
import torch

from oracle import oracle as orc
from random_cases import group_values

pytestmark = pytest.mark.gpu
TD = {orc.DT_BF16: torch.bfloat16, orc.DT_F16: torch.float16, orc.DT_F32: torch.float32}


def ops():
    from auto_round_amd import ops as _ops

    return _ops


@pytest.mark.parametrize("seed", range(24))
def test_int_qdq_forward_backward_on_random_ranges(seed):
    rng = np.random.default_rng(1000 + seed)                       # the draws of the live reference test, case by case
    nbits = int(rng.choice([2, 3, 4, 8]))
    gs = int(rng.choice([16, 32, 64, 128, 40, 256]))
    sym = bool(rng.integers(0, 2))
    w_dt = [orc.DT_BF16, orc.DT_BF16, orc.DT_F16, orc.DT_F32][int(rng.integers(0, 4))]
    s_dt = orc.DT_F16 if rng.integers(0, 4) else orc.DT_F32
    G = int(rng.integers(8, 96))
    thresh = 1e-8 if s_dt == orc.DT_F32 else 1e-5
    W = group_values(rng, G, gs, TD[w_dt])
    if w_dt == orc.DT_F16:
        W = W.clamp(-6e4, 6e4)
    V = torch.from_numpy(((rng.random((G, gs)) - 0.5) * 1.4).astype(np.float32))
    ms = torch.from_numpy(rng.uniform(0.0, 1.0, G).astype(np.float32))
    Ms = torch.from_numpy(rng.uniform(0.0, 1.0, G).astype(np.float32))
    ms[:2], Ms[:2] = 1.0, 1.0
    ms[2], Ms[2] = 0.0, 0.0
    dWq = torch.from_numpy((rng.standard_normal((G, gs)) * 1e-3).astype(np.float32)).to(TD[w_dt])
    dWq.view(-1)[::53] = 0
    o = ops()
    Wb = orc.to_bits(W).reshape(-1)
    omin, omax = orc.group_minmax(Wb, w_dt, G, gs)
    a = (Wb, V.numpy().reshape(-1), omin, omax, ms.numpy(), Ms.numpy(), G, gs, nbits, int(sym), w_dt, s_dt, thresh)
    Wq_o, s_o, zp_o = orc.qdq_int_fwd(*a)
    dV_o, dmin_o, dmax_o = orc.qdq_int_bwd(orc.to_bits(dWq).reshape(-1), *a)
    Wd, Vd, msd, Msd = W.cuda().view(-1), V.cuda().view(-1), ms.cuda(), Ms.cuda()
    wmin, wmax = o.group_minmax(Wd, gs)
    tag = f"bits={nbits} gs={gs} sym={sym} w_dt={w_dt} s_dt={s_dt} G={G}"
    assert np.array_equal(orc.to_bits(wmin), omin) and np.array_equal(orc.to_bits(wmax), omax), tag
    kw = dict(gs=gs, bits=nbits, sym=sym, scale_dtype=TD[s_dt], q_thresh=thresh)
    Wq, s, zp = o.qdq_int_fwd(Wd, Vd, wmin, wmax, msd, Msd, want_scale=True, **kw)
    assert np.array_equal(orc.to_bits(s), s_o) and np.array_equal(zp.cpu().numpy(), zp_o), tag
    # fp16 weights only: where W/scale overflows fp16 the reference's round_ste ((round(x) - x) + x) turns the inf into NaN and
    # poisons the weight; the kernel keeps the saturated code instead (deliberate deviation)
    nan_o = np.isnan(orc.from_bits(Wq_o, TD[w_dt]).float().numpy())
    assert not nan_o.any() or (w_dt == orc.DT_F16 and nan_o.mean() < 0.01), tag
    assert np.array_equal(orc.to_bits(Wq)[~nan_o], Wq_o[~nan_o]) and bool(torch.isfinite(Wq.float()).all()), tag
    dV, dmin, dmax = o.qdq_int_bwd(dWq.cuda().view(-1), Wd, Vd, wmin, wmax, msd, Msd, **kw)
    assert np.array_equal(dV.cpu().numpy().view(np.uint32), dV_o.view(np.uint32)), tag
    for mine, ref in ((dmin.cpu().numpy(), dmin_o), (dmax.cpu().numpy(), dmax_o)):
        ok = np.isfinite(ref)       # fp16 weights: the reference's own autograd overflows to inf/NaN in a few tiny groups
        big = np.abs(ref[ok]).max() if ok.any() else 0.0
        bad = ok & (np.sign(mine) != np.sign(ref))
        assert np.all(np.abs(ref[bad]) <= 1e-4 * big) and bad.mean() < 0.02, (tag, int(bad.sum()))


@pytest.mark.parametrize("seed", range(12))
def test_fp4_qdq_forward_on_random_ranges(seed):
    rng = np.random.default_rng(2000 + seed)
    nv = bool(seed % 2)
    gs = 16 if nv else 32
    w_dt = orc.DT_BF16 if seed % 3 else orc.DT_F16
    G = int(rng.integers(8, 64))
    W = group_values(rng, G, gs, TD[w_dt])
    if w_dt == orc.DT_F16:
        W = W.clamp(-6e4, 6e4)
    V = torch.from_numpy(((rng.random((G, gs)) - 0.5) * 1.4).astype(np.float32))
    Ms = torch.from_numpy(rng.uniform(0.3, 1.0, G).astype(np.float32))
    o = ops()
    Wb = orc.to_bits(W).reshape(-1)
    Wd = W.cuda().view(-1)
    absmax, tmax = o.group_absmax(Wd, gs, want_tensor_max=True)
    if nv:
        gsc_o = orc.nvfp4_global_scale(Wb, w_dt)
        gsc = (448.0 * 6.0 * (1.0 / tmax)).to(torch.float32)
        assert np.float32(gsc.item()) == np.float32(gsc_o)
        ref, sc_o = orc.qdq_nvfp4_fwd(Wb, V.numpy().reshape(-1), Ms.numpy(), gsc_o, G, gs, w_dt)[:2]
        Wq, sc = o.qdq_fp4_fwd(Wd, V.cuda().view(-1), absmax, Ms.cuda(), mode=1, gs=gs, global_scale=gsc, want_scale=True)
        assert np.array_equal(sc.cpu().numpy(), np.asarray(sc_o, np.float32))
    else:
        gsc_o, gsc = 1.0, None
        ref, sc_o = orc.qdq_mxfp4_fwd(Wb, V.numpy().reshape(-1), Ms.numpy(), G, gs, w_dt)[:2]
        Wq, sc = o.qdq_fp4_fwd(Wd, V.cuda().view(-1), absmax, Ms.cuda(), mode=0, gs=gs, want_scale=True)
        assert np.array_equal(orc.to_bits(sc), sc_o)
    assert np.array_equal(orc.to_bits(Wq), ref), f"nv={nv} G={G} w_dt={w_dt}"
    # nibbles + scale bytes of the baked weight (one group per row)
    packed_o, sb_o = orc.pack_fp4(ref, sc_o, G, gs, gs, int(nv), w_dt, global_scale=gsc_o)
    packed, sb = o.pack_fp4(Wq.view(G, gs), sc, mode=int(nv), gs=gs, global_scale=gsc)
    assert np.array_equal(packed.cpu().numpy(), packed_o) and np.array_equal(sb.cpu().numpy(), sb_o.reshape(sb.shape))


@pytest.mark.parametrize("seed", range(16))
def test_int_activation_fake_quant_on_random_ranges(seed):
    rng = np.random.default_rng(4000 + seed)
    nbits = int(rng.choice([4, 8]))
    hidden = int(rng.choice([128, 256, 384]))
    gs = int(rng.choice([32, 128, -1]))
    sym = bool(seed % 2)
    dt = orc.DT_F16 if seed % 4 == 3 else orc.DT_BF16
    g = hidden if gs == -1 else gs
    T = int(rng.integers(3, 24))
    x = group_values(rng, T * hidden // g, g, torch.float32).clamp(-3e4, 3e4).reshape(T, hidden).to(TD[dt])
    dy = torch.from_numpy((rng.standard_normal((T, hidden)) * 1e-2).astype(np.float32)).to(TD[dt])
    xb, dyb, G = orc.to_bits(x).reshape(-1), orc.to_bits(dy).reshape(-1), T * hidden // g
    o = ops()
    tag = f"bits={nbits} gs={gs} hidden={hidden} sym={sym} dt={dt} T={T}"
    if sym:
        xq_o, s_o = orc.int_act_fwd(xb, G, g, nbits, a_dt=dt)
        dx_o = orc.int_act_bwd(dyb, xb, G, g, nbits, a_dt=dt)
    else:
        xq_o, s_o, _ = orc.int_act_asym_fwd(xb, G, g, nbits, a_dt=dt)
        dx_o = orc.int_act_asym_bwd(dyb, xb, G, g, nbits, a_dt=dt)
    xq, s = o.qdq_int_act_fwd(x.cuda().view(-1), gs=g, bits=nbits, sym=sym, want_scale=True)
    dx = o.int_act_bwd(dy.cuda().view(-1), x.cuda().view(-1), gs=g, bits=nbits, sym=sym)
    assert np.array_equal(orc.to_bits(s), s_o), tag
    assert np.array_equal(orc.to_bits(xq), xq_o), tag
    finite = np.isfinite(orc.from_bits(dx_o, TD[dt]).float().numpy())
    assert (orc.to_bits(dx) == dx_o)[finite].mean() >= 0.998, tag


@pytest.mark.parametrize("seed", range(12))
def test_packers_on_random_ranges(seed):
    """Both GPTQ-order conventions and the AWQ container on a random baked layer (the live test's draws): the oracle forward
    bakes the layer, the HIP packers must reproduce the oracle's words, zeros and transposed scales."""
    rng = np.random.default_rng(3000 + seed)
    nbits = int(rng.choice([2, 3, 4, 8]))
    gs = int(rng.choice([32, 64, 128]))
    sym = bool(rng.integers(0, 2))
    out_f, in_f = 32 * int(rng.integers(1, 5)), gs * int(rng.integers(1, 4)) * (2 if gs == 32 else 1)
    if in_f % 32:
        in_f *= 2
    G = out_f * in_f // gs
    W = group_values(rng, G, gs, torch.bfloat16)
    V = (rng.random((G, gs)) - 0.5).astype(np.float32)
    Wb = orc.to_bits(W).reshape(-1)
    wmin, wmax = orc.group_minmax(Wb, orc.DT_BF16, G, gs)
    one = np.ones(G, np.float32)
    Wq_o, s_o, zp_o = orc.qdq_int_fwd(Wb, V.reshape(-1), wmin, wmax, one, one, G, gs, nbits, int(sym))
    zp_arg = float(zp_o[0]) if sym else zp_o.reshape(out_f, -1)
    o = ops()
    Wq = orc.from_bits(Wq_o, torch.bfloat16).cuda().view(out_f, in_f)
    sc = orc.from_bits(s_o, torch.float16).cuda().view(out_f, -1)
    zp = zp_arg if sym else torch.from_numpy(zp_arg).cuda()
    tag = f"bits={nbits} gs={gs} sym={sym} {out_f}x{in_f}"
    for off in (1, 0):
        qw_o, qz_o, st_o = orc.pack_int(Wq_o, s_o, zp_arg, out_f, in_f, gs, nbits, zp_off=off)
        qw, qz, st = o.pack_int(Wq, sc, zp, gs=gs, bits=nbits, zp_off=off)
        assert np.array_equal(qw.cpu().numpy(), qw_o) and np.array_equal(qz.cpu().numpy(), qz_o), (tag, off)
        assert np.array_equal(orc.to_bits(st), st_o), (tag, off)
    if nbits == 4:
        qw_o, qz_o, st_o = orc.pack_awq(Wq_o, s_o, zp_arg, out_f, in_f, gs)
        qw, qz, st = o.pack_awq(Wq, sc, zp, gs=gs)
        assert np.array_equal(qw.cpu().numpy(), qw_o) and np.array_equal(qz.cpu().numpy(), qz_o) and np.array_equal(orc.to_bits(st), st_o), tag
