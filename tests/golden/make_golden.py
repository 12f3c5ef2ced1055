#!/usr/bin/env python
"""Generate the golden fixtures in tests/golden/ by running the REFERENCE itself.

Run in the build container only (needs /root/reference; the GPU box never runs this):

    cd /tmp && PYTHONDONTWRITEBYTECODE=1 python /root/repo/tests/golden/make_golden.py

The reference is imported from /root/reference with the `cpuinfo` shim in oracle/ref_shim
(the only missing dependency).  Everything is seeded; outputs are small .npz
files that are committed.  16-bit floats are stored as uint16 bit patterns.

Fixtures (what each pins, and the reference entry point that produced it):
  int_qdq_<cfg>.npz     quant_tensor_sym / quant_tensor_asym fwd + autograd bwd   (data_type/int.py)
  step_<cfg>.npz        WrapperLinear._qdq_weight -> backward(dWq) -> SignSGD.step x3 with LinearLR
                        (wrapper.py, sign_round/sign_sgd.py)
  pack_int_<cfg>.npz    QuantLinear.pack of qlinear_torch_zp / qlinear_torch (auto_round_extension/torch)
  fp4_<kind>.npz        quant_mx / nv_fp4 fwd + autograd bwd, qlinear_fp pack        (data_type/mxfp.py, nvfp.py)
  known_answers.npz     the reference's own golden vectors (cast_to_fp4 15-value, E2M1 12-value, nibbles)
  sampler.npz           IndexSampler stream after transformers.set_seed(42)          (compressors/utils.py)
  mse.npz               MSELoss fwd + (loss*1000).backward() on bf16 activations     (sign_round/quantizer.py)
"""
import os
import random
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, os.path.join(REPO, "oracle", "ref_shim"))
sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

torch.set_num_threads(4)


def bits(t):
    t = t.detach().cpu().contiguous()
    if t.dtype in (torch.bfloat16, torch.float16):
        return t.view(torch.int16).numpy().view(np.uint16).copy()
    if t.dtype == torch.float8_e4m3fn:
        return t.view(torch.uint8).numpy().copy()
    return t.numpy().copy()


def make_weight(rows, cols, gs, dtype, seed, heavy_tail=True):
    """Weights with the corner cases the kernels must get right baked into specific groups."""
    g = torch.Generator().manual_seed(seed)
    w = torch.randn(rows, cols, generator=g) * 0.02
    if heavy_tail:
        torch.manual_seed(seed)              # StudentT samples from the GLOBAL generator: seed it first (reproducible fixtures)
        t = torch.distributions.StudentT(4.0).sample((rows, cols)) * 0.02
        w = torch.where(torch.rand(rows, cols, generator=g) < 0.3, t, w)
    wg = w.view(-1, gs)
    wg[0] = 0.0                              # all-zero group  -> scale threshold path
    wg[1] = wg[1].abs()                      # all-positive    -> wmin clamps to 0
    wg[2] = -wg[2].abs()                     # all-negative    -> wmax clamps to 0
    wg[3] = wg[3] * 1e-4                     # tiny group      -> |scale| near q_scale_thresh
    wg[4] = wg[4].abs(); wg[4, 0] = -wg[4].max()   # |min| == |max| tie (sym: a == b)
    wg[5] = wg[5] * 1e-7                     # below fp16 threshold
    return w.to(dtype)


INT_CFGS = [
    # name, bits, gs, sym, wdtype, scale dtype
    ("w4g128_sym_bf16", 4, 128, True, torch.bfloat16, torch.float16),
    ("w2g32_asym_bf16", 2, 32, False, torch.bfloat16, torch.float16),
    ("w3g128_sym_bf16", 3, 128, True, torch.bfloat16, torch.float16),
    ("w8g128_sym_bf16", 8, 128, True, torch.bfloat16, torch.float16),
    ("w4g32_asym_bf16", 4, 32, False, torch.bfloat16, torch.float16),
    ("w2g64_sym_bf16", 2, 64, True, torch.bfloat16, torch.float16),
    ("w4g128_sym_f16", 4, 128, True, torch.float16, torch.float16),
    ("w4g128_asym_f32", 4, 128, False, torch.float32, torch.float16),
    ("w4g128_sym_bf16_s32", 4, 128, True, torch.bfloat16, torch.float32),
]


def gen_int_qdq():
    from auto_round.data_type.int import quant_tensor_asym, quant_tensor_sym

    for name, nbits, gs, sym, wdt, sdt in INT_CFGS:
        rows, cols = 48, 256
        W = make_weight(rows, cols, gs, wdt, seed=zlib.crc32(name.encode()) % 1000)
        G = rows * cols // gs
        g = torch.Generator().manual_seed(7)
        V = ((torch.rand(G, gs, generator=g) - 0.5) * 1.2).requires_grad_(True)
        # already-clamped scales (the wrapper clamps before calling the quant function)
        ms = (0.4 + 0.6 * torch.rand(G, generator=g)).clamp(0, 1)
        Ms = (0.4 + 0.6 * torch.rand(G, generator=g)).clamp(0, 1)
        ms[7] = 0.0; Ms[7] = 0.0          # both zero -> scale 0 -> threshold
        ms[8] = 1.0; Ms[8] = 1.0
        ms.requires_grad_(True); Ms.requires_grad_(True)
        Wg = W.view(-1, gs)
        wmin = torch.clamp(Wg.min(1)[0], max=0)
        wmax = torch.clamp(Wg.max(1)[0], min=0)
        thresh = 1e-8 if sdt == torch.float32 else 1e-5
        fn = quant_tensor_sym if sym else quant_tensor_asym
        Wq, scale, zp = fn(W, bits=nbits, group_size=gs, v=V, min_scale=ms, max_scale=Ms, scale_dtype=sdt,
                           tensor_min=wmin, tensor_max=wmax, q_scale_thresh=thresh)
        dWq = (torch.randn(rows, cols, generator=g) * 1e-3).to(wdt)
        dWq.view(-1)[::97] = 0                                 # exact zeros -> sign(0)=0
        Wq.backward(dWq)
        out = dict(W=bits(W), V=V.detach().numpy(), min_scale=ms.detach().numpy(), max_scale=Ms.detach().numpy(),
                   wmin=bits(wmin), wmax=bits(wmax), Wq=bits(Wq), scale=bits(scale.reshape(-1)),
                   dWq=bits(dWq), dV=V.grad.numpy(), dmin=ms.grad.numpy(), dmax=Ms.grad.numpy(),
                   meta=np.array([nbits, gs, int(sym), rows, cols]), thresh=np.float32(thresh))
        out["zp"] = (np.full(G, float(zp), np.float32) if not isinstance(zp, torch.Tensor)
                     else zp.detach().reshape(-1).float().numpy())
        np.savez_compressed(os.path.join(HERE, f"int_qdq_{name}.npz"), **out)
        print("int_qdq", name, "neg-scale groups:", int((scale.float() < 0).sum()), "/", G)


def attr_linear(in_f, out_f, nbits, gs, sym, wdt, data_type="int", act_bits=16, seed=0, scale_dtype=torch.float16):
    lin = torch.nn.Linear(in_f, out_f, bias=False)
    lin.weight.data = make_weight(out_f, in_f, gs, wdt, seed)
    lin.weight.requires_grad_(False)
    for k, v in dict(bits=nbits, group_size=gs, sym=sym, data_type=data_type, scale_dtype=scale_dtype,
                     act_bits=act_bits, act_group_size=gs, act_sym=True, act_dynamic=True,
                     act_data_type=data_type).items():
        setattr(lin, k, v)
    return lin


def gen_steps():
    """Three tuning steps with injected dWq, through the reference wrapper + optimizer + scheduler."""
    from auto_round.algorithms.quantization.sign_round.sign_sgd import SignSGD
    from auto_round.wrapper import WrapperLinear

    for name, nbits, gs, sym, wdt, sdt in INT_CFGS[:4]:
        out_f, in_f, iters = 32, 256, 200
        lin = attr_linear(in_f, out_f, nbits, gs, sym, wdt, seed=11, scale_dtype=sdt)
        W0 = lin.weight.data.clone()
        w = WrapperLinear(lin, enable_minmax_tuning=True, enable_torch_compile=False, device="cpu", iters=iters)
        lr0 = 1.0 / iters
        lr = torch.tensor(lr0)
        opt = SignSGD([{"params": [w.params["value"]], "lr": torch.tensor(lr0)},
                       {"params": [w.params["min_scale"], w.params["max_scale"]], "lr": torch.tensor(lr0)}],
                      lr=lr, weight_decay=0)
        sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=iters)
        g = torch.Generator().manual_seed(3)
        rec = dict(W=bits(W0), wmin=bits(w.weight_min), wmax=bits(w.weight_max),
                   meta=np.array([nbits, gs, int(sym), out_f, in_f, iters]))
        # start from a non-trivial point so clamps are live
        with torch.no_grad():
            w.value.copy_((torch.rand(w.value.shape, generator=g) - 0.5))
            w.min_scale.copy_(0.9 + 0.2 * torch.rand(w.min_scale.shape, generator=g))   # some > 1 -> clamp
            w.max_scale.copy_(0.9 + 0.2 * torch.rand(w.max_scale.shape, generator=g))
        rec["V0"] = w.value.detach().numpy().copy()
        rec["min0"] = w.min_scale.detach().numpy().copy()
        rec["max0"] = w.max_scale.detach().numpy().copy()
        for step in range(3):
            dWq = (torch.randn(out_f, in_f, generator=g) * 1e-3).to(wdt)
            wq, _, _ = w._qdq_weight(w.value, w.min_scale, w.max_scale)
            rec[f"Wq{step}"] = bits(wq)
            wq.backward(dWq)
            rec[f"dWq{step}"] = bits(dWq)
            rec[f"lr{step}"] = np.float32(opt.param_groups[0]["lr"].item())
            rec[f"gmin{step}"] = w.min_scale.grad.numpy().copy()
            rec[f"gmax{step}"] = w.max_scale.grad.numpy().copy()
            opt.step(); opt.zero_grad(); sched.step()
            rec[f"V{step + 1}"] = w.value.detach().numpy().copy()
            rec[f"min{step + 1}"] = w.min_scale.detach().numpy().copy()
            rec[f"max{step + 1}"] = w.max_scale.detach().numpy().copy()
        # unwrapper with the final params: baked weight + scale + zp
        best = {k: v.data.clone() for k, v in w.params.items()}
        with torch.no_grad():
            layer = w.unwrapper(best)
        rec["W_final"] = bits(layer.weight.data)
        rec["scale_final"] = bits(layer.scale)
        rec["zp_final"] = (np.float32(layer.zp) if not isinstance(layer.zp, torch.Tensor)
                           else layer.zp.float().numpy())
        # the whole LinearLR stream, to pin the host-side schedule
        lin2 = torch.nn.Parameter(torch.zeros(1))
        o2 = SignSGD([{"params": [lin2], "lr": torch.tensor(lr0)}], lr=torch.tensor(lr0), weight_decay=0)
        s2 = torch.optim.lr_scheduler.LinearLR(o2, start_factor=1.0, end_factor=0.0, total_iters=iters)
        lrs = []
        for _ in range(iters):
            lrs.append(float(o2.param_groups[0]["lr"]))
            o2.step(); s2.step()
        rec["lr_stream"] = np.array(lrs, dtype=np.float32)
        np.savez_compressed(os.path.join(HERE, f"step_{name}.npz"), **rec)
        print("step", name)


def gen_pack_int():
    from auto_round.data_type.int import quant_tensor_asym, quant_tensor_sym
    import auto_round_extension.torch.qlinear_torch as plain
    import auto_round_extension.torch.qlinear_torch_zp as zpmod

    cfgs = [("w4g128_sym", 4, 128, True), ("w2g32_sym", 2, 32, True), ("w8g128_sym", 8, 128, True),
            ("w3g128_sym", 3, 128, True), ("w4g128_asym", 4, 128, False), ("w2g32_asym", 2, 32, False),
            ("w3g128_asym", 3, 128, False), ("w8g64_asym", 8, 64, False)]
    for name, nbits, gs, sym in cfgs:
        out_f, in_f = 64, 256
        lin = attr_linear(in_f, out_f, nbits, gs, sym, torch.bfloat16, seed=5)
        g = torch.Generator().manual_seed(9)
        V = (torch.rand(out_f * in_f // gs, gs, generator=g) - 0.5)
        fn = quant_tensor_sym if sym else quant_tensor_asym
        Wq, scale, zp = fn(lin.weight.data, bits=nbits, group_size=gs, v=V)
        lin.weight.data.copy_(Wq)
        scale2d = scale.reshape(out_f, -1)
        zp2d = zp.reshape(out_f, -1) if isinstance(zp, torch.Tensor) else zp
        rec = dict(Wq=bits(lin.weight.data), scale=bits(scale2d), meta=np.array([nbits, gs, int(sym), out_f, in_f]))
        rec["zp"] = zp2d.float().numpy() if isinstance(zp2d, torch.Tensor) else np.float32(zp2d)
        for tag, mod in (("zp", zpmod), ("plain", plain)):
            ql = mod.QuantLinear(nbits, gs, in_f, out_f, False)
            ql.device = "cpu"   # export.pack_layer sets this before pack() (export_to_autoround/export.py:211)
            z = zp2d.clone() if isinstance(zp2d, torch.Tensor) else zp2d
            ql.pack(lin, scale2d.clone(), z, None, device="cpu")
            rec[f"qweight_{tag}"] = ql.qweight.numpy().copy()
            rec[f"qzeros_{tag}"] = ql.qzeros.numpy().copy()
            rec[f"scales_{tag}"] = bits(ql.scales)
        if nbits == 4:   # AWQ GEMM container through the reference's own from_linear (export_to_awq/export.py:129-142)
            from auto_round.export.export_to_awq.utils import WQLinear_GEMM

            z_awq = zp2d.t().contiguous().to(torch.float32) if isinstance(zp2d, torch.Tensor) else zp2d
            aq = WQLinear_GEMM.from_linear(lin, nbits, gs, scales=scale2d.t().contiguous(), zeros=z_awq, device="cpu")
            rec.update(awq_qweight=aq.qweight.numpy().copy(), awq_qzeros=aq.qzeros.numpy().copy(), awq_scales=bits(aq.scales))
        np.savez_compressed(os.path.join(HERE, f"pack_int_{name}.npz"), **rec)
        print("pack_int", name, rec["qweight_zp"].shape, rec["qzeros_zp"].shape)


def gen_fp4():
    from auto_round.data_type.mxfp import quant_element, quant_mx
    from auto_round.data_type.nvfp import calculate_gparam, cast_to_fp4, nv_fp4
    from auto_round.export.export_to_autoround.qlinear_fp import QuantLinear as FpQL
    from auto_round.export.export_to_autoround.qlinear_fp import _pack_fp4_to_uint8

    rows, cols = 32, 256
    for kind, gs in (("mxfp4", 32), ("nvfp4", 16)):
        W = make_weight(rows, cols, gs, torch.bfloat16, seed=21)
        G = rows * cols // gs
        g = torch.Generator().manual_seed(17)
        V = ((torch.rand(G, gs, generator=g) - 0.5)).requires_grad_(True)
        Ms = (0.5 + 0.5 * torch.rand(G, generator=g)).requires_grad_(True)
        rec = dict(W=bits(W), V=V.detach().numpy(), max_scale=Ms.detach().numpy(), meta=np.array([gs, rows, cols]))
        if kind == "mxfp4":
            Wq, se, _ = quant_mx(W, bits=4, group_size=gs, v=V, max_scale=Ms, data_type="mx_fp")
            rec["exp"] = bits(se.reshape(-1))
        else:
            gsc = calculate_gparam(W, gs)
            Wq, sc, _ = nv_fp4(W, bits=4, group_size=gs, v=V, global_scale=gsc, max_scale=Ms)
            rec["scale"] = sc.detach().reshape(-1).float().numpy()
            rec["global_scale"] = np.float32(gsc.item())
        dWq = (torch.randn(rows, cols, generator=g) * 1e-3).to(torch.bfloat16)
        Wq.backward(dWq)
        rec.update(Wq=bits(Wq), dWq=bits(dWq), dV=V.grad.numpy(), dmax=Ms.grad.numpy())
        # pack through the reference QuantLinear
        lin = torch.nn.Linear(cols, rows, bias=False)
        lin.weight.data = Wq.detach().clone()
        if kind == "mxfp4":
            ql = FpQL(4, gs, cols, rows, False, data_type="mx_fp", act_bits=16)
            ql.pack(lin, se.detach().reshape(rows, -1), device="cpu")
        else:
            ql = FpQL(4, gs, cols, rows, False, data_type="nv_fp", act_bits=16)
            ql.pack(lin, sc.detach().reshape(rows, -1), global_scale=gsc, device="cpu")
            rec["weight_global_scale"] = ql.weight_global_scale.numpy().copy()
        rec["weight_packed"] = ql.weight_packed.numpy().copy()
        rec["weight_scale_bytes"] = bits(ql.weight_scale) if ql.weight_scale.dtype != torch.uint8 else ql.weight_scale.numpy().copy()
        np.savez_compressed(os.path.join(HERE, f"fp4_{kind}.npz"), **rec)
        print("fp4", kind, "packed", rec["weight_packed"].shape, ql.weight_scale.dtype)

    # the reference's own known-answer vectors
    d12 = torch.tensor([0.0, 0.25, 0.4, 0.75, 1.25, 1.4, 1.75, 2.5, 2.9, 3.5, 5.0, 5.1])
    d15 = torch.tensor([0.0, 0.25, 0.4, 0.75, 1.25, 1.4, 1.75, 2.5, 2.9, 3.5, 5.0, 5.1, 6.0, 6.2, 8.9])
    gt12 = torch.tensor([0.0, 0.0, 0.5, 1.0, 1.0, 1.5, 2.0, 2.0, 3.0, 4.0, 4.0, 6.0])            # mxfp.py:417-425
    gt15 = torch.tensor([0.0, 0.0, 0.5, 1.0, 1.0, 1.5, 2.0, 2.0, 3.0, 4.0, 4.0, 6.0, 6.0, 6.0, 6.0])  # test_nvfp.py:75-80
    assert torch.equal(quant_element(d12, 2, 3, 6.0), gt12)
    assert torch.equal(cast_to_fp4(d15), gt15)
    dense = torch.linspace(-6.5, 6.5, 2081)
    x4 = torch.tensor([[0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0], [-0.0, -0.5, -1.0, -1.5, -2.0, -3.0, -4.0, -6.0],
                       [6.0] * 8, [100.0] * 8, [-6.0] * 8, [0.5] * 8, [0.2, 0.3, 0.74, 0.76, 1.24, 1.26, 2.4, 2.6]])
    np.savez_compressed(
        os.path.join(HERE, "known_answers.npz"),
        e2m1_in12=d12.numpy(), e2m1_gt12=gt12.numpy(), fp4_in15=d15.numpy(), fp4_gt15=gt15.numpy(),
        dense_in=dense.numpy(), dense_cast_to_fp4=cast_to_fp4(dense).numpy(),
        dense_quant_element=quant_element(dense.clamp(-6, 6), 2, 3, 6.0).numpy(),
        nib_in=x4.numpy(), nib_out=_pack_fp4_to_uint8(x4).numpy(),
        e4m3_in=torch.linspace(-448, 448, 4001).numpy(),
        e4m3_bits=bits(torch.linspace(-448, 448, 4001).to(torch.float8_e4m3fn)),
        e4m3_small_in=torch.logspace(-12, 2, 1500, base=2.0).numpy(),
        e4m3_small_bits=bits(torch.logspace(-12, 2, 1500, base=2.0).to(torch.float8_e4m3fn)),
    )
    print("known answers ok")


def gen_sampler():
    from transformers import set_seed

    from auto_round.compressors.utils import IndexSampler

    rec = {}
    for nsamples, bs, iters in ((128, 8, 200), (16, 4, 30), (10, 3, 25)):
        set_seed(42)
        s = IndexSampler(nsamples, bs)
        rec[f"n{nsamples}_b{bs}"] = np.array([s.next_batch() for _ in range(iters)], dtype=np.int64)
        # a second block continues the same global stream (what sharded execution must replay)
        s2 = IndexSampler(nsamples, bs)
        rec[f"n{nsamples}_b{bs}_block2"] = np.array([s2.next_batch() for _ in range(iters)], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "sampler.npz"), **rec)
    print("sampler ok")


def gen_mse():
    g = torch.Generator().manual_seed(5)
    pred = (torch.randn(2, 16, 64, generator=g)).to(torch.bfloat16).requires_grad_(True)
    ref = (pred.detach().float() + 0.05 * torch.randn(2, 16, 64, generator=g)).to(torch.bfloat16)
    loss = torch.nn.MSELoss()(pred.to(torch.float32), ref.to(torch.float32))
    (loss * 1000).backward()
    rec = dict(pred=bits(pred), ref=bits(ref), loss=np.float32(loss.item()), dpred=bits(pred.grad))
    # the same through the reference's own masked loss (SignRoundQuantizer._get_loss with a valid-token mask)
    from types import SimpleNamespace

    from auto_round.algorithms.quantization.sign_round.quantizer import SignRoundQuantizer as RefQ

    fake = SimpleNamespace(model_context=SimpleNamespace(amp=True, amp_dtype=torch.bfloat16))
    mask = [torch.ones(1, 16, dtype=torch.long), torch.ones(1, 16, dtype=torch.long)]
    mask[0][0, -1] = 0; mask[1][0, -1] = 0; mask[1][0, 3:6] = 0
    pred2 = pred.detach().clone().requires_grad_(True)
    loss2 = RefQ._get_loss(fake, pred2, ref, [0, 1], torch.nn.MSELoss(), "cpu", mask)
    (loss2 * 1000).backward()
    rec.update(mask=torch.cat(mask).reshape(-1).numpy().astype(np.uint8), loss_masked=np.float32(loss2.item()),
               dpred_masked=bits(pred2.grad))
    np.savez_compressed(os.path.join(HERE, "mse.npz"), **rec)
    print("mse ok")


if __name__ == "__main__":
    random.seed(0); torch.manual_seed(0)
    which = sys.argv[1:] or ["int", "step", "pack", "fp4", "sampler", "mse"]
    if "int" in which: gen_int_qdq()
    if "step" in which: gen_steps()
    if "pack" in which: gen_pack_int()
    if "fp4" in which: gen_fp4()
    if "sampler" in which: gen_sampler()
    if "mse" in which: gen_mse()


def gen_act():
    """Activation fake-quant (dynamic MXFP4; NVFP4 with a static global scale) forward + autograd w.r.t. the input:
    WrapperLinear._qdq_act (wrapper.py:295-321) -> quant_mx / nv_fp4_with_static_gs with v=0, max_scale=1."""
    from auto_round.data_type.mxfp import quant_mx
    from auto_round.data_type.nvfp import nv_fp4_with_static_gs

    g = torch.Generator().manual_seed(31)
    x = torch.randn(4, 24, 128, generator=g)
    x[0, 0, :32] = 0.0                       # an all-zero group
    x[0, 1, :32] = x[0, 1, :32].abs()
    x[0, 2, 5] = 40.0                        # an outlier: most of its group rounds to 0
    x[0, 3, 7] = -x[0, 3, :32].abs().max() * 1.0   # tie in |x| -> first index wins the max gradient
    x[0, 3, 3] = x[0, 3, 7].abs()
    for kind in ("mxfp4", "nvfp4"):
        xb = x.to(torch.bfloat16).requires_grad_(True)
        if kind == "mxfp4":
            xq, _, _ = quant_mx(xb, bits=4, group_size=32, v=0, max_scale=torch.tensor(1.0), data_type="mx_fp")
            rec = {}
        else:
            act_max = xb.detach().float().abs().max()
            xq, _, _ = nv_fp4_with_static_gs(xb, bits=4, group_size=16, v=0, tensor_max=act_max)
            rec = {"act_max": np.float32(act_max.item())}
        dy = (torch.randn(x.shape, generator=g) * 1e-2).to(torch.bfloat16)
        xq.backward(dy)
        rec.update(x=bits(xb), xq=bits(xq), dy=bits(dy), dx=bits(xb.grad))
        np.savez_compressed(os.path.join(HERE, f"act_{kind}.npz"), **rec)
        print("act", kind, "nan grads:", int(torch.isnan(xb.grad.float()).sum()))


if __name__ == "__main__" and ("act" in sys.argv[1:] or not sys.argv[1:]):
    gen_act()


def gen_search():
    """Init-scale searches of the algorithm extension: search_mx_scale / search_nvfp4_scale / search_scales (+ the
    threshold clamp of search_int), with a per-input-channel importance matrix expanded like reshape_imatrix_for_weight."""
    from auto_round.data_type.int import search_scales
    from auto_round.data_type.mxfp import search_mx_scale
    from auto_round.data_type.nvfp import search_nvfp4_scale
    from auto_round.data_type.utils import reshape_imatrix_for_weight, search_optimized_init_scale

    rows, cols = 48, 256
    g = torch.Generator().manual_seed(77)
    imatrix = (torch.rand(cols, generator=g) * 4.0 + 0.01) ** 2 * 100.0
    imatrix[5] = 0.0
    rec = {"imatrix": imatrix.numpy()}
    for kind, gs in (("mxfp4", 32), ("nvfp4", 16)):
        W = make_weight(rows, cols, gs, torch.bfloat16, seed=33)
        Wg = W.reshape(-1, gs)
        qw = reshape_imatrix_for_weight(imatrix, Wg, gs)
        fn = search_mx_scale if kind == "mxfp4" else search_nvfp4_scale
        best_qw = fn(Wg, 4, qw)
        best_1 = fn(Wg, 4, torch.ones_like(Wg, dtype=torch.float32))
        rec[f"{kind}_W"] = bits(W)
        rec[f"{kind}_best_qw"] = best_qw.reshape(-1).float().numpy()
        rec[f"{kind}_best_ones"] = best_1.reshape(-1).float().numpy()
        print("search", kind, "distinct:", sorted(set(np.round(best_qw.reshape(-1).float().numpy(), 2).tolist()))[:8])
    for nbits, gs in ((4, 128), (2, 32), (3, 64)):
        W = make_weight(rows, cols, gs, torch.bfloat16, seed=34 + nbits)
        Wg = W.reshape(-1, gs)
        qw = reshape_imatrix_for_weight(imatrix, Wg, gs)
        s_raw = search_scales(Wg, nbits, qw)
        s_init = search_optimized_init_scale(Wg, "int_sym", nbits, qw, 1e-5)
        s_ones = search_optimized_init_scale(Wg, "int_sym", nbits, None, 1e-5)
        rec[f"int{nbits}g{gs}_W"] = bits(W)
        rec[f"int{nbits}g{gs}_raw"] = bits(s_raw.reshape(-1))
        rec[f"int{nbits}g{gs}_init"] = bits(s_init.reshape(-1))
        rec[f"int{nbits}g{gs}_init_ones"] = bits(s_ones.reshape(-1))
        print("search int", nbits, gs, s_init.dtype, tuple(s_init.shape))
    np.savez_compressed(os.path.join(HERE, "search.npz"), **rec)


if __name__ == "__main__" and ("search" in sys.argv[1:] or not sys.argv[1:]):
    gen_search()


def gen_outlier_loss():
    """Outlier-suppressed loss of the algorithm extension through SignRoundV2Quantizer._get_loss."""
    from types import SimpleNamespace

    from auto_round.algorithms.quantization.sign_roundv2.quantizer import SignRoundV2Quantizer as V2

    g = torch.Generator().manual_seed(13)
    pred = torch.randn(4, 64, 128, generator=g).to(torch.bfloat16)
    ref = (pred.float() + 0.05 * torch.randn(4, 64, 128, generator=g)).to(torch.bfloat16)
    ref.view(-1)[::997] += 3.0            # a few real outliers
    fake = SimpleNamespace(_use_outlier_suppressed_loss=True, amp=True, amp_dtype=torch.bfloat16)
    rec = dict(pred=bits(pred), ref=bits(ref))
    for tag, vmask in (("plain", None), ("masked", [torch.ones(1, 64, dtype=torch.long) for _ in range(4)])):
        if vmask is not None:
            for m in vmask:
                m[0, -1] = 0
        p2 = pred.detach().clone().requires_grad_(True)
        loss = V2._get_loss(fake, p2, ref, [0, 1, 2, 3], torch.nn.MSELoss(), "cpu", vmask)
        (loss * 1000).backward()
        rec[f"loss_{tag}"] = np.float32(loss.item())
        rec[f"dpred_{tag}"] = bits(p2.grad)
        if vmask is not None:
            rec["mask"] = torch.cat(vmask).reshape(-1).numpy().astype(np.uint8)
    np.savez_compressed(os.path.join(HERE, "outlier_loss.npz"), **rec)
    print("outlier loss ok", rec["loss_plain"], rec["loss_masked"])


if __name__ == "__main__" and ("outlier" in sys.argv[1:] or not sys.argv[1:]):
    gen_outlier_loss()


def gen_steps_v2():
    """Three tuning steps through the algorithm extension's SignRoundOptimizedWrapperLinear (searched init_scale from the
    layer's imatrix, min/max-scale bounds (0, 2)) + the reference optimizer, for the three data types it supports."""
    from auto_round.algorithms.quantization.sign_round.sign_sgd import SignSGD
    from auto_round.algorithms.quantization.sign_roundv2.quantizer import SignRoundOptimizedWrapperLinear as OptW

    out_f, in_f, iters = 32, 256, 200
    g0 = torch.Generator().manual_seed(99)
    imatrix = ((torch.rand(in_f, generator=g0) * 3.0 + 0.05) ** 2 * 50.0).to(torch.float32)
    for name, nbits, gs, dtype_name in (("w4g128", 4, 128, "int"), ("w2g32", 2, 32, "int"), ("mxfp4", 4, 32, "mx_fp"),
                                        ("nvfp4", 4, 16, "nv_fp")):
        lin = attr_linear(in_f, out_f, nbits, gs, True, torch.bfloat16, data_type=dtype_name, seed=41)
        lin.imatrix = imatrix.clone()
        lin.iters = iters
        W0 = lin.weight.data.clone()
        w = OptW(lin, enable_minmax_tuning=True, enable_torch_compile=False, device="cpu")
        lr0 = 1.0 / iters
        opt = SignSGD([{"params": [w.params["value"]], "lr": torch.tensor(lr0)},
                       {"params": [w.params["min_scale"], w.params["max_scale"]], "lr": torch.tensor(lr0)}],
                      lr=torch.tensor(lr0), weight_decay=0)
        sched = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=iters)
        g = torch.Generator().manual_seed(4)
        init = w.init_scale
        rec = dict(W=bits(W0), imatrix=imatrix.numpy(), meta=np.array([nbits, gs, out_f, in_f, iters]),
                   init_scale=bits(init.reshape(-1)) if init.dtype != torch.float32 else init.reshape(-1).numpy())
        if dtype_name == "nv_fp":
            rec["global_scale"] = np.float32(w.weight_global_scale.item())
        with torch.no_grad():
            w.value.copy_((torch.rand(w.value.shape, generator=g) - 0.5))
            w.max_scale.copy_(0.7 + 1.4 * torch.rand(w.max_scale.shape, generator=g))   # some > 2 -> the (0,2) clamp is live
        rec["V0"] = w.value.detach().numpy().copy()
        rec["max0"] = w.max_scale.detach().numpy().copy()
        for step in range(3):
            dWq = (torch.randn(out_f, in_f, generator=g) * 1e-3).to(torch.bfloat16)
            wq, _, _ = w._qdq_weight(w.value, w.min_scale, w.max_scale)
            rec[f"Wq{step}"] = bits(wq)
            wq.backward(dWq)
            rec[f"dWq{step}"] = bits(dWq)
            rec[f"gmax{step}"] = w.max_scale.grad.numpy().copy()
            assert w.min_scale.grad is None
            opt.step(); opt.zero_grad(); sched.step()
            rec[f"V{step + 1}"] = w.value.detach().numpy().copy()
            rec[f"max{step + 1}"] = w.max_scale.detach().numpy().copy()
        best = {k: v.data.clone() for k, v in w.params.items()}
        with torch.no_grad():
            layer = w.unwrapper(best)
        rec["W_final"] = bits(layer.weight.data)
        sc = layer.scale
        rec["scale_final"] = bits(sc.reshape(-1)) if sc.dtype in (torch.float16, torch.bfloat16, torch.uint8) else sc.reshape(-1).float().numpy()
        np.savez_compressed(os.path.join(HERE, f"stepv2_{name}.npz"), **rec)
        print("stepv2", name, "init dtype", init.dtype, tuple(init.shape), "scale", sc.dtype, tuple(sc.shape))


if __name__ == "__main__" and ("stepv2" in sys.argv[1:] or not sys.argv[1:]):
    gen_steps_v2()


def gen_int_act():
    """Dynamic symmetric INT activation fake-quant (W4A8-style schemes): quant_tensor_sym on the activation with v=0 and
    the wrapper's non-tunable 0-dim act_min_scale / act_max_scale, forward + autograd backward w.r.t. the input."""
    from auto_round.data_type.int import quant_tensor_asym, quant_tensor_sym

    rec = {}
    for tag, nbits, gs, dt, hidden in (("a8g32", 8, 32, torch.bfloat16, 128), ("a8g128", 8, 128, torch.bfloat16, 256),
                                       ("a4g32", 4, 32, torch.bfloat16, 128), ("a8g32_f16", 8, 32, torch.float16, 128),
                                       ("a8pt", 8, -1, torch.bfloat16, 256), ("asym_a8g32", 8, 32, torch.bfloat16, 128),
                                       ("asym_a4g128", 4, 128, torch.bfloat16, 256), ("asym_a8pt_f16", 8, -1, torch.float16, 256)):
        g = torch.Generator().manual_seed(31 + nbits + max(gs, 0))
        x = (torch.randn(3, 7, hidden, generator=g) * 1.7)
        x[0, 0, :32] = 0.0                       # all-zero group
        x[0, 1, :32] = x[0, 1, :32].abs()        # all-positive group
        x[0, 2, :32] = -x[0, 2, :32].abs()       # all-negative group
        x[0, 3, 5] = 9.0; x[0, 3, 9] = 9.0       # tied maxima
        x[0, 4, 3] = -11.0; x[0, 4, 4] = 11.0    # |min| == max
        x[1, 0, :32] *= 1e-7                     # below the scale threshold
        x = x.to(dt).requires_grad_(True)
        one = torch.tensor(1.0)
        fn = quant_tensor_asym if tag.startswith("asym") else quant_tensor_sym
        xq, scale, zp = fn(x, bits=nbits, group_size=gs, v=0, min_scale=one.clone(), max_scale=one.clone(),
                           scale_dtype=torch.float16, tensor_max=None, q_scale_thresh=1e-5)
        if tag.startswith("asym"):
            rec[f"{tag}_zp"] = zp.detach().reshape(-1).float().numpy()
        dy = (torch.randn(x.shape, generator=g) * 1e-2).to(dt)
        xq.backward(dy)
        rec.update({f"{tag}_x": bits(x), f"{tag}_xq": bits(xq), f"{tag}_scale": bits(scale.reshape(-1)), f"{tag}_dy": bits(dy),
                    f"{tag}_dx": bits(x.grad), f"{tag}_meta": np.array([nbits, gs, hidden])})
        print("int act", tag, xq.dtype, scale.dtype, tuple(scale.shape))
    np.savez_compressed(os.path.join(HERE, "int_act.npz"), **rec)


if __name__ == "__main__" and ("intact" in sys.argv[1:] or not sys.argv[1:]):
    gen_int_act()
