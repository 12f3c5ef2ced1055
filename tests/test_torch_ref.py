"""Pins oracle/torch_ref.py (the loop-level checker and the bench's CPU baseline):
  * bit-for-bit against the committed golden vectors the reference produced (always runs),
  * bit-for-bit against the reference's own wrapper_block + SignSGD + LinearLR loop on a seeded OPT-shaped decoder
    layer, when /root/reference is importable (build container only; the GPU box skips it)."""
import glob
import os
import random
import sys

import numpy as np
import pytest
import torch

from oracle import oracle as orc
from oracle import torch_ref as tr

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
TDT = {orc.DT_BF16: torch.bfloat16, orc.DT_F16: torch.float16, orc.DT_F32: torch.float32}
REF = "/root/reference"


def cfg_dtypes(name):
    w_dt = orc.DT_F16 if name.endswith("_f16") else orc.DT_F32 if name.endswith("_f32") else orc.DT_BF16
    s_dt = orc.DT_F32 if name.endswith("_s32") else orc.DT_F16
    return w_dt, s_dt


INT_FILES = sorted(glob.glob(os.path.join(GOLDEN, "int_qdq_*.npz")))


@pytest.mark.parametrize("path", INT_FILES, ids=[os.path.basename(p)[8:-4] for p in INT_FILES])
def test_qdq_int_forward_backward_equals_reference_golden(path):
    z = np.load(path)
    nbits, gs, sym, rows, cols = [int(x) for x in z["meta"]]
    w_dt, s_dt = cfg_dtypes(os.path.basename(path)[8:-4])
    W = orc.from_bits(z["W"], w_dt).reshape(rows, cols)
    V = torch.from_numpy(z["V"].copy()).requires_grad_(True)
    ms = torch.from_numpy(z["min_scale"].copy()).requires_grad_(True)
    Ms = torch.from_numpy(z["max_scale"].copy()).requires_grad_(True)
    Wq, s, zp = tr.qdq_int(W, nbits, gs, bool(sym), V, ms, Ms, orc.from_bits(z["wmin"], w_dt), orc.from_bits(z["wmax"], w_dt),
                           TDT[s_dt], float(z["thresh"]))
    assert np.array_equal(orc.to_bits(Wq), z["Wq"])
    assert np.array_equal(orc.to_bits(s.reshape(-1)), z["scale"])
    Wq.backward(orc.from_bits(z["dWq"], w_dt).reshape(rows, cols))
    assert np.array_equal(V.grad.numpy(), z["dV"])
    assert np.array_equal(ms.grad.numpy(), z["dmin"]) and np.array_equal(Ms.grad.numpy(), z["dmax"])


def test_linear_lr_stream_matches_reference_schedule():
    z = np.load(os.path.join(GOLDEN, "step_w4g128_sym_bf16.npz"))
    assert np.array_equal(np.array(tr.linear_lr_stream(1.0 / 200, 200), dtype=np.float32), z["lr_stream"])


def test_sampler_stream_matches_reference_index_sampler():
    z = np.load(os.path.join(GOLDEN, "sampler.npz"))
    from auto_round_amd.quantizer import IndexSampler

    for cls in (tr.Sampler, IndexSampler):
        for nsamples, bs, iters in ((128, 8, 200), (16, 4, 30), (10, 3, 25)):
            random.seed(42)   # transformers.set_seed(42) seeds `random` with the same value
            s = cls(nsamples, bs)
            got = np.array([s.next_batch() for _ in range(iters)])
            assert np.array_equal(got, z[f"n{nsamples}_b{bs}"]), cls
            s2 = cls(nsamples, bs)
            got2 = np.array([s2.next_batch() for _ in range(iters)])
            assert np.array_equal(got2, z[f"n{nsamples}_b{bs}_block2"]), cls


def _opt_layer(seed=0, hidden=64, ffn=128, heads=4, group_size=32, bits=4, sym=True):
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer

    torch.manual_seed(seed)
    cfg = OPTConfig(hidden_size=hidden, ffn_dim=ffn, num_attention_heads=heads, num_hidden_layers=1, vocab_size=128,
                    max_position_embeddings=64, word_embed_proj_dim=hidden)
    cfg._attn_implementation = "eager"
    layer = OPTDecoderLayer(cfg).to(torch.bfloat16).eval()
    for p in layer.parameters():
        p.requires_grad_(False)
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            for k, v in dict(bits=bits, group_size=group_size, sym=sym, data_type="int", scale_dtype=torch.float16, act_bits=16,
                             act_group_size=32, act_sym=True, act_dynamic=True, act_data_type="int").items():
                setattr(m, k, v)
    return layer


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("group_size,bits,sym", [(32, 4, True), (48, 4, True), (-1, 4, False), (40, 2, False), (0, 4, True), (0, 8, False)])
def test_tune_block_equals_reference_loop_on_cpu(group_size, bits, sym):
    """Same seeded layer, same data, 6 iterations: reference wrapper_block/SignSGD/LinearLR vs oracle/torch_ref."""
    import copy

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.algorithms.quantization.sign_round.sign_sgd import SignSGD
    from auto_round.compressors.utils import IndexSampler, collect_best_params
    from auto_round.wrapper import unwrapper_block, wrapper_block

    iters, bs, N, S, H = 6, 2, 8, 16, 64
    g = torch.Generator().manual_seed(3)
    X = torch.randn(N, S, H, generator=g).to(torch.bfloat16)
    base = _opt_layer(group_size=group_size, bits=bits, sym=sym)   # 48 / 40 do not divide 64 or 128: padded groups
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        Y = torch.cat([base(X[i:i + 1])[0] if isinstance(base(X[i:i + 1]), tuple) else base(X[i:i + 1]) for i in range(N)])

    def fwd(blk, x, others):
        out = blk(x)
        return out[0] if isinstance(out, tuple) else out

    # --- reference
    blk_ref = copy.deepcopy(base)
    random.seed(7)
    wrapper_block(blk_ref, True, False, enable_torch_compile=False, device="cpu")
    wr = {n: m for n, m in blk_ref.named_modules() if hasattr(m, "orig_layer")}
    rp = [m.params["value"] for m in wr.values()]
    mp = [p for m in wr.values() for k, p in m.params.items() if "min" in k or "max" in k]
    lr0 = 1.0 / iters
    opt = SignSGD([{"params": rp, "lr": torch.tensor(lr0)}, {"params": mp, "lr": torch.tensor(lr0)}], lr=torch.tensor(lr0),
                  weight_decay=0)
    sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=iters)
    sampler = IndexSampler(N, bs)
    best_loss, best = float("inf"), {}
    ref_losses = []
    for i in range(iters):
        idx = sampler.next_batch()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = fwd(blk_ref, X[idx], {})
        loss = torch.nn.MSELoss()(out.float(), Y[idx].float())
        ref_losses.append(loss.item())
        (loss * 1000).backward()
        if loss.item() < best_loss:
            best_loss = loss.item()
            best = collect_best_params(blk_ref, "cpu")
        opt.step(); opt.zero_grad(); sch.step()
    with torch.no_grad():
        unwrapper_block(blk_ref, best)

    # --- oracle restatement
    blk_o = copy.deepcopy(base)
    random.seed(7)
    best_o, info = tr.tune_block(blk_o, X, Y, {}, iters=iters, batch_size=bs, forward=fwd)
    assert info["losses"] == ref_losses
    for (n1, m1), (n2, m2) in zip(blk_ref.named_modules(), blk_o.named_modules()):
        if isinstance(m1, torch.nn.Linear):
            assert torch.equal(m1.weight.view(torch.int16), m2.weight.view(torch.int16)), n1
            assert torch.equal(m1.scale.view(torch.int16), m2.scale.view(torch.int16)), n1
            assert torch.equal(torch.as_tensor(m1.zp), torch.as_tensor(m2.zp))
    for n in best:
        for k in best[n]:
            assert torch.equal(best[n][k], best_o[n][k]), (n, k)


def test_fp4_restatements_equal_reference_golden():
    """qdq_mxfp4 / qdq_nvfp4 (weights, with V and max_scale) and the activation variants: forward bits and autograd
    gradients identical to the reference's (same torch ops on CPU -> same bits, NaNs included)."""
    for kind, gs in (("mxfp4", 32), ("nvfp4", 16)):
        z = np.load(os.path.join(GOLDEN, f"fp4_{kind}.npz"))
        _, rows, cols = [int(x) for x in z["meta"]]
        W = orc.from_bits(z["W"], orc.DT_BF16).reshape(rows, cols)
        V = torch.from_numpy(z["V"].copy()).requires_grad_(True)
        Ms = torch.from_numpy(z["max_scale"].copy()).requires_grad_(True)
        if kind == "mxfp4":
            Wq, se = tr.qdq_mxfp4(W, gs, V, Ms)
            assert np.array_equal(orc.to_bits(se.reshape(-1)), z["exp"])
        else:
            gsc = tr.nvfp4_global_scale(W)
            assert np.float32(gsc.item()) == z["global_scale"]
            Wq, sc = tr.qdq_nvfp4(W, gs, V, Ms, gsc)
            assert np.array_equal(sc.detach().reshape(-1).numpy(), z["scale"])
        assert np.array_equal(orc.to_bits(Wq), z["Wq"])
        Wq.backward(orc.from_bits(z["dWq"], orc.DT_BF16).reshape(rows, cols))
        assert np.array_equal(V.grad.numpy(), z["dV"])
        assert np.array_equal(Ms.grad.numpy(), z["dmax"], equal_nan=True)

        za = np.load(os.path.join(GOLDEN, f"act_{kind}.npz"))
        x = orc.from_bits(za["x"], orc.DT_BF16).requires_grad_(True)

        class L:  # the attributes act_fake_quant reads
            act_data_type = "mx_fp" if kind == "mxfp4" else "nv_fp4_with_static_gs"
            act_group_size = gs
            act_max = None if kind == "mxfp4" else float(za["act_max"])

        xq = tr.act_fake_quant(x, L)
        assert np.array_equal(orc.to_bits(xq), za["xq"])
        xq.backward(orc.from_bits(za["dy"], orc.DT_BF16))
        a, b = orc.to_bits(x.grad), za["dx"]
        nan = np.isnan(orc.from_bits(b, orc.DT_BF16).float().numpy())
        assert np.array_equal(a[~nan], b[~nan])


# ---- algorithm extension (SignRoundV2) -------------------------------------------------------------------------------
def test_alg_ext_init_scale_searches_equal_reference_golden():
    z = np.load(os.path.join(GOLDEN, "search.npz"))
    im = torch.from_numpy(z["imatrix"].astype(np.float32))
    for nb, gs in ((4, 128), (2, 32), (3, 64)):
        Wg = orc.from_bits(z[f"int{nb}g{gs}_W"].reshape(-1), orc.DT_BF16).reshape(-1, gs)
        qw = im.reshape(1, -1).expand(Wg.numel() // im.numel(), -1).reshape(Wg.shape)
        assert np.array_equal(orc.to_bits(tr.search_int_scale(Wg, nb, qw).reshape(-1)), z[f"int{nb}g{gs}_init"])
        assert np.array_equal(orc.to_bits(tr.search_int_scale(Wg, nb, None).reshape(-1)), z[f"int{nb}g{gs}_init_ones"])
    for kind, gs, fn in (("mxfp4", 32, tr.search_mx_coeff), ("nvfp4", 16, tr.search_nv_coeff)):
        Wg = orc.from_bits(z[f"{kind}_W"].reshape(-1), orc.DT_BF16).reshape(-1, gs)
        qw = im.reshape(1, -1).expand(Wg.numel() // im.numel(), -1).reshape(Wg.shape)
        assert np.array_equal(fn(Wg, qw).reshape(-1).numpy(), z[f"{kind}_best_qw"])
        assert np.array_equal(fn(Wg, None).reshape(-1).numpy(), z[f"{kind}_best_ones"])


V2_FILES = sorted(glob.glob(os.path.join(GOLDEN, "stepv2_*.npz")))


@pytest.mark.parametrize("path", V2_FILES, ids=[os.path.basename(p)[7:-4] for p in V2_FILES])
def test_alg_ext_wrapper_steps_equal_reference_golden(path):
    """RefOptWrapperLinear == the reference's SignRoundOptimizedWrapperLinear: init scale, three fwd/bwd/sign-SGD steps."""
    z = np.load(path)
    nbits, gs, out_f, in_f, iters = [int(x) for x in z["meta"]]
    kind = os.path.basename(path)[7:-4]
    lin = torch.nn.Linear(in_f, out_f, bias=False)
    lin.weight.data = orc.from_bits(z["W"], orc.DT_BF16).reshape(out_f, in_f)
    lin.weight.requires_grad_(False)
    dt = {"mxfp4": "mx_fp", "nvfp4": "nv_fp"}.get(kind, "int")
    for k, v in dict(bits=nbits, group_size=gs, sym=True, data_type=dt, scale_dtype=torch.float16, act_bits=16).items():
        setattr(lin, k, v)
    lin.imatrix = torch.from_numpy(z["imatrix"].copy())
    w = tr.RefOptWrapperLinear(lin)
    init = w.init_scale.reshape(-1)
    if kind.startswith("w"):
        assert np.array_equal(orc.to_bits(init), z["init_scale"])
    else:
        assert np.array_equal(init.numpy(), z["init_scale"])
    with torch.no_grad():
        w.value.copy_(torch.from_numpy(z["V0"].copy()))
        w.max_scale.copy_(torch.from_numpy(z["max0"].copy()))
    lrs = tr.linear_lr_stream(1.0 / iters, iters)
    for step in range(3):
        wq, _, _ = w.qdq()
        assert np.array_equal(orc.to_bits(wq), z[f"Wq{step}"]), f"step {step}: Wq"
        wq.backward(orc.from_bits(z[f"dWq{step}"], orc.DT_BF16).reshape(out_f, in_f))
        assert w.min_scale.grad is None
        assert np.array_equal(w.max_scale.grad.numpy(), z[f"gmax{step}"], equal_nan=True), f"step {step}: d max_scale"   # all-zero fp4 group: NaN in both
        tr.sign_sgd_step(list(w.params.values()), lrs[step])
        for p in w.params.values():
            p.grad = None
        assert np.array_equal(w.value.detach().numpy(), z[f"V{step + 1}"]), f"step {step}: V"
        assert np.array_equal(w.max_scale.detach().numpy(), z[f"max{step + 1}"], equal_nan=True), f"step {step}: max_scale"
    layer = w.unwrap({k: p.data.clone() for k, p in w.params.items()})
    assert np.array_equal(orc.to_bits(layer.weight.data), z["W_final"])


def test_alg_ext_outlier_loss_equals_reference_golden():
    z = np.load(os.path.join(GOLDEN, "outlier_loss.npz"))
    pred = orc.from_bits(z["pred"], orc.DT_BF16).reshape(4, 64, 128)
    ref = orc.from_bits(z["ref"], orc.DT_BF16).reshape(4, 64, 128)
    for tag, m in (("plain", None), ("masked", torch.from_numpy(z["mask"].astype(np.int64)).reshape(4, 64, 1))):
        p = pred.clone().requires_grad_(True)
        loss = tr.outlier_loss(p, ref, m)
        (loss * 1000).backward()
        assert loss.item() == float(z[f"loss_{tag}"])
        assert np.array_equal(orc.to_bits(p.grad), z[f"dpred_{tag}"])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("bits,data_type,group_size", [(4, "int", 32), (2, "int", 32), (4, "mx_fp", 32), (4, "nv_fp", 16)])
def test_alg_ext_tune_block_equals_reference_loop_on_cpu(bits, data_type, group_size):
    """Same seeded layer and data, 5 iterations of the reference's V2 pieces (imatrix hooks, optimized wrapper, V2
    loss, SignSGD, LinearLR) vs tune_block(alg_ext=True)."""
    import copy
    from types import SimpleNamespace

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.algorithms.quantization.sign_round.sign_sgd import SignSGD
    from auto_round.algorithms.quantization.sign_roundv2.quantizer import SignRoundOptimizedWrapperLinear, SignRoundV2Quantizer
    from auto_round.compressors.utils import IndexSampler, collect_best_params
    from auto_round.wrapper import unwrapper_block, wrapper_block

    iters, bs, N, S, H = 5, 2, 8, 16, 64
    g = torch.Generator().manual_seed(5)
    X = torch.randn(N, S, H, generator=g).to(torch.bfloat16)
    base = _opt_layer(group_size=group_size, bits=bits, sym=True)
    for m in base.modules():
        if isinstance(m, torch.nn.Linear):
            m.data_type = data_type
            m.act_data_type = data_type
    def fwd(blk, x, others):
        out = blk(x)
        return out[0] if isinstance(out, tuple) else out

    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        Y = torch.cat([fwd(base, X[i:i + 1], {}) for i in range(N)])
    use_outlier = bits < 4

    # --- reference pieces
    blk_ref = copy.deepcopy(base)
    fake = SimpleNamespace(_use_outlier_suppressed_loss=use_outlier, amp=True, amp_dtype=torch.bfloat16,
                           _is_wint4aint4=lambda: False)
    hs = SignRoundV2Quantizer._register_imatrix_hooks(fake, blk_ref)
    with torch.no_grad(), torch.autocast("cpu", dtype=torch.bfloat16):
        for b0 in range(0, N, bs):
            fwd(blk_ref, X[b0:b0 + bs], {})
    for h in hs:
        h.remove()
    random.seed(7)
    wrapper_block(blk_ref, True, False, enable_torch_compile=False, device="cpu", wrapper_cls=SignRoundOptimizedWrapperLinear)
    wr = {n: m for n, m in blk_ref.named_modules() if hasattr(m, "orig_layer")}
    rp = [m.params["value"] for m in wr.values()]
    mp = [p for m in wr.values() for k, p in m.params.items() if "min" in k or "max" in k]
    lr0 = 1.0 / iters
    opt = SignSGD([{"params": rp, "lr": torch.tensor(lr0)}, {"params": mp, "lr": torch.tensor(lr0)}], lr=torch.tensor(lr0),
                  weight_decay=0)
    sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=iters)
    sampler = IndexSampler(N, bs)
    best_loss, best, ref_losses = float("inf"), {}, []
    for i in range(iters):
        idx = sampler.next_batch()
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = fwd(blk_ref, X[idx], {})
        if use_outlier:
            loss = SignRoundV2Quantizer._get_loss(fake, out, Y[idx], idx, torch.nn.MSELoss(), "cpu", None)
        else:
            loss = torch.nn.MSELoss()(out.float(), Y[idx].float())
        ref_losses.append(loss.item())
        (loss * 1000).backward()
        if loss.item() < best_loss:
            best_loss = loss.item()
            best = collect_best_params(blk_ref, "cpu")
        opt.step(); opt.zero_grad(); sch.step()
    with torch.no_grad():
        unwrapper_block(blk_ref, best)

    # --- restatement
    blk_o = copy.deepcopy(base)
    tr.collect_imatrix(blk_o, X, {}, batch_size=bs, forward=fwd)
    random.seed(7)
    best_o, info = tr.tune_block(blk_o, X, Y, {}, iters=iters, batch_size=bs, forward=fwd, alg_ext=True)
    assert info["losses"] == ref_losses
    for (n1, m1), (n2, m2) in zip(blk_ref.named_modules(), blk_o.named_modules()):
        if isinstance(m1, torch.nn.Linear):
            assert torch.equal(m1.weight.view(torch.int16), m2.weight.view(torch.int16)), n1
    for n in best:
        for k in best[n]:
            assert torch.equal(best[n][k], best_o[n][k]), (n, k)


@pytest.mark.parametrize("tag", ("a8g32", "a8g128", "a4g32", "a8g32_f16", "a8pt"))
def test_int_activation_restatement_equals_reference_golden(tag):
    z = np.load(os.path.join(GOLDEN, "int_act.npz"))
    nb, gs, hidden = [int(v) for v in z[tag + "_meta"]]
    dt = orc.DT_F16 if tag.endswith("f16") else orc.DT_BF16
    x = orc.from_bits(z[tag + "_x"], dt).requires_grad_(True)
    xq, s = tr.qdq_int_act_sym(x, nb, gs)
    xq.backward(orc.from_bits(z[tag + "_dy"], dt))
    assert np.array_equal(orc.to_bits(xq), z[tag + "_xq"]) and np.array_equal(orc.to_bits(s.reshape(-1)), z[tag + "_scale"])
    assert np.array_equal(orc.to_bits(x.grad), z[tag + "_dx"])


@pytest.mark.parametrize("tag", ("asym_a8g32", "asym_a4g128", "asym_a8pt_f16"))
def test_asymmetric_int_activation_restatement_equals_reference_golden(tag):
    z = np.load(os.path.join(GOLDEN, "int_act.npz"))
    nb, gs, hidden = [int(v) for v in z[tag + "_meta"]]
    dt = orc.DT_F16 if tag.endswith("f16") else orc.DT_BF16
    x = orc.from_bits(z[tag + "_x"], dt).requires_grad_(True)
    xq, s, zp = tr.qdq_int_act_asym(x, nb, gs)
    xq.backward(orc.from_bits(z[tag + "_dy"], dt))
    assert np.array_equal(orc.to_bits(xq), z[tag + "_xq"]) and np.array_equal(zp.detach().reshape(-1).numpy(), z[tag + "_zp"])
    assert np.array_equal(orc.to_bits(x.grad), z[tag + "_dx"])


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
def test_reference_wrapper_cannot_take_2d_block_groups_for_int_schemes():
    """SURVEY 8 a1 lists tuple (2-D block) group sizes among the reshapes of data_type/utils.py:29-71.  They exist for the FP8 block
    schemes: through WrapperLinear the reference's own INT quant functions fail on them (weight_min is reduced over both block
    dimensions, wrapper.py:154-164, the quant function broadcasts it against a last-dimension reduction) -- which is why the
    MI355X wrapper refuses them with NotImplementedError instead of inventing a behaviour."""
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.wrapper import WrapperLinear

    lin = torch.nn.Linear(128, 64, bias=False).to(torch.bfloat16)
    lin.bits, lin.group_size, lin.sym, lin.data_type, lin.scale_dtype, lin.act_bits, lin.iters = 4, (16, 32), True, "int", torch.float16, 16, 200
    lin.super_bits = lin.super_group_size = None
    w = WrapperLinear(lin, enable_minmax_tuning=True, device="cpu")
    with pytest.raises(RuntimeError):
        w(torch.randn(2, 128).to(torch.bfloat16))
