"""torch restatement of the reference's tuning loop (WrapperLinear fake-quant + autograd + SignSGD + best-MSE
bookkeeping), device agnostic.

TEST INFRASTRUCTURE ONLY (see oracle/ar_oracle.c header): used by tests/ as the loop-level checker, by
__graft_entry__.smoke(), and by bench.py's `cpu_baseline` leg (kind "port": this file timed on the host cores).
Nothing under auto_round_amd/ imports it.

Parity status: PINNED -- tests/test_torch_ref.py checks it bit-for-bit against the committed golden vectors
(tests/golden/int_qdq_*.npz, step_*.npz) that the real reference produced, and, when /root/reference is importable
(build container), directly against the reference's own WrapperLinear / SignSGD on the same seeded layer.

Reference entry points restated here (paths relative to /root/reference):
  qdq_int            <- auto_round/data_type/int.py:165-238 (sym), :241-298 (asym)
  RefWrapperLinear   <- auto_round/wrapper.py:139-293, :517-565
  sign_sgd_step      <- auto_round/algorithms/quantization/sign_round/sign_sgd.py:356-389
  tune_block         <- auto_round/algorithms/quantization/sign_round/quantizer.py:311-552
Algorithm extension (enable_alg_ext / SignRoundV2), pinned by tests/golden/search.npz, stepv2_*.npz, outlier_loss.npz:
  search_int_scale / search_mx_coeff / search_nv_coeff <- data_type/int.py:24-86, mxfp.py:103-170, nvfp.py:329-386
  RefOptWrapperLinear <- algorithms/quantization/sign_roundv2/quantizer.py:101-161
  outlier_loss        <- algorithms/quantization/sign_roundv2/quantizer.py:362-399
  collect_imatrix     <- algorithms/quantization/sign_roundv2/quantizer.py:413-428
"""
from __future__ import annotations

import random
from typing import Dict, List, Optional

import torch
import torch.nn.functional as F


def _ste(fn, x):
    return (fn(x) - x).detach() + x


def qdq_int(W2d, bits, gs, sym, v, min_scale, max_scale, wmin, wmax, scale_dtype=torch.float16, thresh=1e-5):
    """Fake-quant of a [out, in] weight with per-group scales; differentiable w.r.t. v, min_scale, max_scale.
    Returns (Wq [out,in] in W dtype, scale [G,1], zp (int for sym, [G,1] tensor for asym)).
    reference: auto_round/data_type/int.py:165-238 (quant_tensor_sym), :241-298 (quant_tensor_asym)"""
    out_f, in_f = W2d.shape
    if gs == 0:                 # per-tensor: ONE group = the whole weight (data_type/utils.py:57-59)
        pad, Wg = 0, W2d.reshape(1, -1)
    else:
        pad = (-in_f) % gs      # zero-pad every row to a multiple of gs (data_type/utils.py:52-56), cut again at the end
        Wg = (F.pad(W2d, (0, pad)) if pad else W2d).reshape(-1, gs)

    def back(x):
        x = x.to(W2d.dtype)
        return x.reshape(out_f, in_f + pad)[:, :in_f] if pad else x.reshape(W2d.shape)

    if sym:
        maxq = 2 ** (bits - 1)
        a = -(wmin * min_scale)
        b = wmax * max_scale
        sgn = 2 * (b < a).int() - 1
        s = ((sgn * torch.max(b, a)) / maxq).to(scale_dtype)
        s = torch.where(s < 0, torch.clamp(s, max=-thresh), torch.clamp(s, min=thresh)).unsqueeze(-1)
        q = torch.clamp(_ste(torch.round, Wg / s + v), -maxq, maxq - 1)
        return back(s * q), s, maxq
    maxq = 2 ** bits - 1
    lo = wmin * min_scale
    hi = wmax * max_scale
    s = torch.clamp(((hi - lo) / maxq).to(scale_dtype), min=thresh)
    zp = _ste(torch.round, -lo / s).unsqueeze(-1)
    s = s.unsqueeze(-1)
    q = torch.clamp(_ste(torch.round, Wg / s + v) + zp, 0, maxq)
    return back(s * (q - zp)), s, zp


def _recip0(x):
    return torch.where(x == 0, torch.zeros_like(x), 1.0 / x)


def _e2m1_mx(t):
    """E2M1 element rounding with the MX reference's STE structure (data_type/mxfp.py:49-85, ebits=2 mbits=3)."""
    pe = _ste(torch.floor, torch.log2(t.abs() + (t == 0).to(t.dtype))).clip(min=0.0)
    x = t / (2.0 ** pe.float()) * 2.0
    a = x.abs()
    tie = ((a - 0.5) % 2 == torch.zeros_like(a)).to(t.dtype)
    x = torch.sign(x) * (_ste(torch.floor, a + 0.5) - tie)
    x = x / 2.0 * (2.0 ** pe.float())
    return torch.clamp(x, min=-6.0, max=6.0)


def qdq_mxfp4(X, gs, v, max_scale, init_scale=1.0):
    """MXFP4 fake-quant (data_type/mxfp.py:233-291).  X [..., in]; groups of gs along the last dim.
    Returns (Xq like X, shared exponent [G,1] in X dtype)."""
    od = X.dtype
    t = X.reshape(-1, gs).to(torch.float32)
    m, _ = torch.max(t.abs(), dim=-1, keepdim=True)
    m = m * (init_scale * max_scale.unsqueeze(-1))
    se = torch.where(m == 0, torch.ones_like(m), torch.log2(m))
    se = (_ste(torch.floor, se) - 2).clamp(min=-127.0, max=127.0)
    sc = torch.pow(2.0, se.float())
    t = torch.clamp(t / sc + v, min=-6.0, max=6.0)
    out = _e2m1_mx(t) * sc
    return out.reshape(X.shape).to(od), se.to(od)


def _e2m1_nv(x):
    """cast_to_fp4 (data_type/nvfp.py:26-39)."""
    sign = torch.sign(x)
    a = x.abs()
    s1 = _ste(torch.round, 2.0 * a) / 2.0
    s2 = _ste(torch.round, a)
    s3 = 2.0 * _ste(torch.round, a / 2.0)
    m1, m2 = a < 2.0, a < 4.0
    y = s1 * m1 + s2 * (~m1) * m2 + s3 * (~m1) * (~m2)
    return y.clamp(-6, 6) * sign


def qdq_nvfp4(X, gs, v, max_scale, global_scale, init_scale=1.0):
    """NVFP4 fake-quant (data_type/nvfp.py:67-98).  Returns (Xq like X, e4m3-valued scale [G,1] fp32)."""
    od = X.dtype
    xg = X.reshape(-1, gs)
    if isinstance(max_scale, torch.Tensor) and max_scale.dim() == 1:
        max_scale = max_scale.view(-1, 1)
    coeff = max_scale * init_scale
    vm = torch.max(xg.abs(), dim=-1, keepdim=True)[0].to(torch.float32) * coeff
    s = global_scale * (vm * (1.0 / 6.0))
    s = torch.clamp(s, min=-448.0, max=448.0)
    s = ((s.to(torch.float8_e4m3fn).to(s.dtype) - s).detach() + s).to(torch.float32)
    osc = _recip0(s * _recip0(global_scale))
    x = torch.clamp(xg.to(torch.float32) * osc + v, -6.0, 6.0)
    out = _e2m1_nv(x) * _recip0(osc)
    return out.reshape(X.shape).to(od), s


# ---- algorithm extension: init-scale searches -------------------------------------------------------------------------
def _recip_eps(t):
    """get_reciprocal (utils/common.py:903-920): 1/t in t's dtype, 0 where |t| < eps (1e-5 for fp16, else 1e-30)."""
    eps = 1e-5 if t.dtype == torch.float16 else 1e-30
    ok = t.abs() >= eps
    return torch.where(ok, 1.0 / torch.where(ok, t, torch.ones_like(t)), torch.zeros_like(t))


def int_search_grid(bits, ratio=0.75):
    """reference: the candidate grid of search_scales, auto_round/data_type/int.py:40-59"""
    nmax = 2 ** (bits - 1)
    if bits == 2:
        n, step = 90, 0.01
    else:
        half = nmax * ratio
        step = half / 200 * 2
        n = int(half / step)
    return nmax, [nmax - step * i for i in range(-n, n + 1) if i != 0]


def search_int_scale(Wg, bits, qw=None, thresh=1e-5):
    """Per-group symmetric scale minimising the importance-weighted rounding error over the reference's grid of
    numerators around nmax; everything in Wg's dtype.  Returns the clamped init scale [G, 1].
    reference: search_scales + search_int, auto_round/data_type/int.py:24-86, data_type/utils.py:203-209"""
    nmax, grid = int_search_grid(bits)
    idx = Wg.abs().argmax(dim=-1, keepdim=True)
    rg = _recip_eps(torch.take_along_dim(Wg, idx, dim=-1))

    def trial(c):
        isc = -c * rg
        sc = _recip_eps(isc)
        L = torch.round(isc * Wg).clamp_(-nmax, nmax - 1)
        err = ((sc * L - Wg).to(torch.float32)) ** 2
        if qw is not None:
            err = err * qw
        return sc, err.sum(dim=-1)

    best_s, best_l = trial(nmax)
    for c in grid:
        sc, l = trial(c)
        better = l < best_l
        best_s = torch.where(better.unsqueeze(-1), sc, best_s)
        best_l = torch.where(better, l, best_l)
    return torch.where(best_s < 0, torch.clamp(best_s, max=-thresh), torch.clamp(best_s, min=thresh))


def _search_coeff(qdq_at, X32, cands, qw):
    best_c, best_l = None, None
    for c in cands:
        err = (qdq_at(c) - X32) ** 2
        if qw is not None:
            err = err * qw
        l = err.sum(dim=-1)
        if best_l is None:
            best_l, best_c = l, torch.full_like(l, c)
        else:
            better = l < best_l
            best_l = torch.where(better, l, best_l)
            best_c = torch.where(better, torch.full_like(l, c), best_c)
    return best_c.unsqueeze(-1)


def search_mx_coeff(Wg, qw=None):
    """MXFP4: which of the coefficients 1, 0.5, 2 on the group max gives the smallest weighted error -> [G, 1] fp32.
    reference: search_mx_scale, auto_round/data_type/mxfp.py:103-170"""
    X32 = Wg.to(torch.float32)
    ones = torch.ones(Wg.shape[0], device=Wg.device)
    return _search_coeff(lambda c: qdq_mxfp4(X32, Wg.shape[-1], 0, ones * c)[0], X32, (1.0, 0.5, 2.0), qw)


def search_nv_coeff(Wg, qw=None):
    """NVFP4: coefficient in {1.0, 0.50 .. 1.51} on the group max (global scale = the tensor's own) -> [G, 1] fp32.
    reference: search_nvfp4_scale, auto_round/data_type/nvfp.py:329-386"""
    X32 = Wg.to(torch.float32)
    gsc = nvfp4_global_scale(X32)
    ones = torch.ones(Wg.shape[0], device=Wg.device)
    cands = [1.0] + [v / 100.0 for v in range(50, 152) if v != 100]
    return _search_coeff(lambda c: qdq_nvfp4(X32, Wg.shape[-1], 0, ones * c, gsc)[0], X32, cands, qw)


def outlier_loss(pred, ref, token_mask=None):
    """mean(((|pred-ref| in fp32) * token_mask * keep)^2), keep = all but the max(1, n/1000) largest |pred-ref| ranked in
    the activation dtype (ties at the k-th value: whatever torch.topk picks).
    reference: SignRoundV2Quantizer._get_loss, algorithms/quantization/sign_roundv2/quantizer.py:362-399"""
    d = (pred - ref).abs().view(-1)
    k = max(1, int(d.numel() / 1000))
    keep = torch.ones_like(d, dtype=torch.bool)
    keep[torch.topk(d, k)[1]] = False
    e = (pred.to(torch.float32) - ref.to(torch.float32)).abs() * keep.view_as(pred)
    if token_mask is not None:
        e = e * token_mask
    return torch.mean(e ** 2)


@torch.no_grad()
def collect_imatrix(block, inputs, input_others, batch_size=8, forward=None, amp_dtype=torch.bfloat16, amp=True):
    """Run the fp block over all samples and leave `imatrix` = sum over tokens of x^2 (fp32 [in]) on every layer that
    will be quantised.
    reference: the imatrix forward hooks, algorithms/quantization/sign_roundv2/quantizer.py:401-428"""
    def hook(m, inp, out):
        x = inp[0] if isinstance(inp, (tuple, list)) else inp
        sq = (x.reshape(-1, x.shape[-1]).to(torch.float32) ** 2).sum(dim=0)
        m.imatrix = sq if not hasattr(m, "imatrix") else m.imatrix + sq

    hs = [m.register_forward_hook(hook) for m in block.modules()
          if isinstance(m, torch.nn.Linear) and int(getattr(m, "bits", 16)) < 16]
    for b0 in range(0, inputs.shape[0], batch_size):
        with torch.autocast(device_type=inputs.device.type, dtype=amp_dtype, enabled=amp):
            forward(block, inputs[b0:b0 + batch_size], input_others) if forward else block(inputs[b0:b0 + batch_size], **input_others)
    for h in hs:
        h.remove()


def nvfp4_global_scale(t):
    """reference: calculate_gparam, auto_round/data_type/nvfp.py:56-64"""
    return 448.0 * 6.0 * _recip0(t.to(torch.float32).abs().max())


def qdq_int_act_sym(x, bits, gs, scale_dtype=torch.float16, thresh=1e-5):
    """Dynamic symmetric INT fake-quant of an activation (quant_tensor_sym with v=0, tensor_min/max=None and the wrapper's
    non-tunable 0-dim act_min_scale / act_max_scale; data_type/int.py:165-238, wrapper.py:295-321).  Unlike the weight path
    the range arithmetic stays in the ACTIVATION dtype: a 0-dim fp32 tensor does not promote a bf16 tensor."""
    gs = x.shape[-1] if (gs == -1 or x.shape[-1] < gs) else gs
    t = x.reshape(-1, gs)
    maxq = 2 ** (bits - 1)
    one = torch.tensor(1.0, device=x.device)
    wmin = torch.clamp(t.min(-1)[0], max=0)
    wmax = torch.clamp(t.max(-1)[0], min=0)
    a = -(wmin * one)
    b = wmax * one
    max_v = (2 * (b < a).int() - 1) * torch.max(b, a)
    s = (max_v / maxq).to(scale_dtype)
    s = torch.where(s < 0, torch.clamp(s, max=-thresh), torch.clamp(s, min=thresh)).unsqueeze(-1)
    q = torch.clamp(_ste(torch.round, t / s + 0), -maxq, maxq - 1)
    return (s * q).to(x.dtype).reshape(x.shape), s


def qdq_int_act_asym(x, bits, gs, scale_dtype=torch.float16, thresh=1e-5):
    """Dynamic asymmetric INT fake-quant of an activation (quant_tensor_asym with v=0, tensor_min/max=None, 0-dim min/max
    scale; data_type/int.py:241-298): range arithmetic in the activation dtype, zero point in fp32."""
    gs = x.shape[-1] if (gs == -1 or x.shape[-1] < gs) else gs
    t = x.reshape(-1, gs)
    maxq = 2 ** bits - 1
    one = torch.tensor(1.0, device=x.device)
    wmin = torch.clamp(t.min(-1)[0], max=0) * one
    wmax = torch.clamp(t.max(-1)[0], min=0) * one
    s = torch.clamp(((wmax - wmin) / maxq).to(scale_dtype), min=thresh)
    zp = _ste(torch.round, -wmin / s).unsqueeze(-1)
    s = s.unsqueeze(-1)
    q = torch.clamp(_ste(torch.round, t / s + 0) + zp, 0, maxq)
    return (s * (q - zp)).to(x.dtype).reshape(x.shape), s, zp


def act_fake_quant(x, layer, use_act_max=True):
    """WrapperLinear._qdq_act for the fp4 and dynamic int activation schemes (wrapper.py:295-321)."""
    adt = str(getattr(layer, "act_data_type", ""))
    one = torch.tensor(1.0, device=x.device)
    if adt.startswith("int"):
        if not bool(getattr(layer, "act_sym", True)):
            return qdq_int_act_asym(x, int(layer.act_bits), int(layer.act_group_size), getattr(layer, "scale_dtype", torch.float16))[0]
        return qdq_int_act_sym(x, int(layer.act_bits), int(layer.act_group_size), getattr(layer, "scale_dtype", torch.float16))[0]
    if adt.startswith("mx_fp"):
        return qdq_mxfp4(x, int(layer.act_group_size), 0, one)[0]
    if adt.startswith("nv_fp"):
        am = getattr(layer, "act_max", None) if use_act_max else None
        tmax = x.to(torch.float32).abs().max() if am is None else torch.as_tensor(am, dtype=torch.float32, device=x.device).abs().max()
        return qdq_nvfp4(x, int(layer.act_group_size), 0, 1.0, 448.0 * 6.0 * _recip0(tmax))[0]
    raise NotImplementedError(adt)


def _is_conv1d(m) -> bool:
    try:
        from transformers.pytorch_utils import Conv1D
    except Exception:  # pragma: no cover
        return False
    return isinstance(m, Conv1D)


class RefWrapperLinear(torch.nn.Module):
    """Plain-torch tuning wrapper around an nn.Linear carrying bits/group_size/sym/scale_dtype attributes.
    reference: WrapperLinear, auto_round/wrapper.py:139-293 (parameters, _qdq_weight), :517-565 (forward), :345-468 (unwrapper)"""

    def __init__(self, layer: torch.nn.Linear, enable_minmax_tuning=True):
        super().__init__()
        self.orig_layer = layer
        self.conv1d = _is_conv1d(layer)          # transformers' Conv1D keeps its weight as [in, out] (wrapper.py:150-151)
        self.bits, self.sym = int(layer.bits), bool(layer.sym)
        self.data_type = str(getattr(layer, "data_type", "int"))
        self.act_quant = int(getattr(layer, "act_bits", 16)) <= 8
        gs = int(layer.group_size)
        W = layer.weight.data.t() if self.conv1d else layer.weight.data
        in_features = W.shape[1]
        self.gs = gs if gs == 0 else (in_features if (gs == -1 or in_features < gs) else gs)
        self.scale_dtype = getattr(layer, "scale_dtype", torch.float16)
        self.thresh = 1e-8 if self.scale_dtype == torch.float32 else 1e-5
        pad = 0 if self.gs == 0 else (-W.shape[1]) % self.gs
        Wg = W.reshape(1, -1) if self.gs == 0 else (F.pad(W, (0, pad)) if pad else W).reshape(-1, self.gs)
        self.wmin = torch.clamp(Wg.min(1)[0], max=0)
        self.wmax = torch.clamp(Wg.max(1)[0], min=0)
        dev = W.device
        self.value = torch.nn.Parameter(torch.zeros(Wg.shape, dtype=torch.float32, device=dev))
        self.min_scale = torch.nn.Parameter(torch.ones(Wg.shape[0], dtype=torch.float32, device=dev),
                                            requires_grad=enable_minmax_tuning)
        self.max_scale = torch.nn.Parameter(torch.ones(Wg.shape[0], dtype=torch.float32, device=dev),
                                            requires_grad=enable_minmax_tuning)
        self.params = {"value": self.value}
        if enable_minmax_tuning:
            self.params.update(min_scale=self.min_scale, max_scale=self.max_scale)

    def qdq(self, v=None, mn=None, mx=None):
        v = self.value if v is None else v
        mn = self.min_scale if mn is None else mn
        mx = self.max_scale if mx is None else mx
        mn.data.clamp_(0, 1)
        mx.data.clamp_(0, 1)
        if self.conv1d:          # quantise the [out, in] view, hand back the stored orientation (wrapper.py:263-264, :291-292)
            wq, s, zp = qdq_int(self.orig_layer.weight.t(), self.bits, self.gs, self.sym, v, mn, mx, self.wmin, self.wmax,
                                self.scale_dtype, self.thresh)
            return wq.t(), s, zp
        W = self.orig_layer.weight
        if self.data_type.startswith("mx_fp"):
            wq, se = qdq_mxfp4(W, self.gs, v, mx)
            return wq, se, None
        if self.data_type.startswith("nv_fp"):
            if not hasattr(self, "gscale"):
                self.gscale = getattr(self.orig_layer, "weight_global_scale", None)
                if self.gscale is None:
                    self.gscale = nvfp4_global_scale(W)
                self.gscale = self.gscale.to(W.device)
            wq, sc = qdq_nvfp4(W, self.gs, v, mx, self.gscale)
            return wq, sc, None
        return qdq_int(self.orig_layer.weight, self.bits, self.gs, self.sym, v, mn, mx, self.wmin, self.wmax,
                       self.scale_dtype, self.thresh)

    def forward(self, x):
        wq, _, _ = self.qdq()
        if self.act_quant:
            x = act_fake_quant(x, self.orig_layer)
        if self.conv1d:          # conv1d_forward (wrapper.py:501-515)
            out = torch.addmm(self.orig_layer.bias, x.view(-1, x.size(-1)), wq)
            return out.view(*x.size()[:-1], self.orig_layer.nf)
        return F.linear(x, wq, self.orig_layer.bias)

    def unwrap(self, best: Optional[Dict[str, torch.Tensor]]):
        best = best or {}
        dev = self.orig_layer.weight.device
        v = best.get("value", torch.tensor(0.0)).to(dev)
        mn = best.get("min_scale", torch.tensor(1.0)).to(dev)
        mx = best.get("max_scale", torch.tensor(1.0)).to(dev)
        with torch.no_grad():
            wq, s, zp = self.qdq(v, mn, mx)
            self.orig_layer.weight.data.copy_(wq)
        out_f = self.orig_layer.weight.shape[1] if self.conv1d else self.orig_layer.weight.shape[0]
        self.orig_layer.scale = (s.reshape(out_f, -1) if s.numel() > 1 else s.view(-1)).cpu()      # wrapper.py:393-398
        self.orig_layer.zp = ((zp.reshape(out_f, -1) if zp.numel() > 1 else zp.view(-1)).cpu()) if isinstance(zp, torch.Tensor) else zp
        if self.act_quant:
            return RefWALayer(self.orig_layer)
        return self.orig_layer


class RefOptWrapperLinear(RefWrapperLinear):
    """Algorithm-extension wrapper: searched init scale (weighted by the layer's `imatrix`, consumed here), max_scale
    tunes a coefficient on it within (0, 2), min_scale takes no part.  Symmetric int / mx_fp4 / nv_fp4.
    reference: SignRoundOptimizedWrapperLinear, algorithms/quantization/sign_roundv2/quantizer.py:101-161"""

    bounds = (0.0, 2.0)

    def __init__(self, layer, enable_minmax_tuning=True):
        super().__init__(layer, enable_minmax_tuning)
        W = layer.weight.data
        pad = (-W.shape[1]) % self.gs
        Wg = (F.pad(W, (0, pad)) if pad else W).reshape(-1, self.gs)
        im = getattr(layer, "imatrix", None)
        qw = None
        if isinstance(im, torch.Tensor):
            row = F.pad(im.reshape(1, -1).to(torch.float32), (0, pad), value=1e-5)
            qw = row.expand(W.shape[0], -1).reshape(Wg.shape).to(W.device)
            del layer.imatrix
        if self.data_type.startswith("mx_fp"):
            self.init_scale = search_mx_coeff(Wg, qw)
        elif self.data_type.startswith("nv_fp"):
            self.init_scale = search_nv_coeff(Wg, qw)
        elif self.sym:
            self.init_scale = search_int_scale(Wg, self.bits, qw, self.thresh)
        else:
            raise ValueError("the optimized path needs a symmetric int / mx / nv data type")

    def qdq(self, v=None, mn=None, mx=None):
        v = self.value if v is None else v
        mn = self.min_scale if mn is None else mn
        mx = self.max_scale if mx is None else mx
        mn.data.clamp_(*self.bounds)
        mx.data.clamp_(*self.bounds)
        W = self.orig_layer.weight
        if self.data_type.startswith("mx_fp"):
            wq, se = qdq_mxfp4(W, self.gs, v, mx, self.init_scale)
            return wq, se, None
        if self.data_type.startswith("nv_fp"):
            if not hasattr(self, "gscale"):
                g = getattr(self.orig_layer, "weight_global_scale", None)
                self.gscale = (nvfp4_global_scale(W) if g is None else g).to(W.device)
            wq, sc = qdq_nvfp4(W, self.gs, v, mx, self.gscale, self.init_scale)
            return wq, sc, None
        out_f, in_f = W.shape
        pad = (-in_f) % self.gs
        Wg = (F.pad(W, (0, pad)) if pad else W).reshape(-1, self.gs)
        maxq = 2 ** (self.bits - 1)
        s = (self.init_scale * (mx.unsqueeze(-1) if mx.dim() == 1 else mx)).to(self.scale_dtype)
        s = torch.where(s < 0, torch.clamp(s, max=-self.thresh), torch.clamp(s, min=self.thresh))
        q = torch.clamp(_ste(torch.round, Wg / s + v), -maxq, maxq - 1)
        wq = (s * q).to(W.dtype).reshape(out_f, in_f + pad)[:, :in_f]
        return wq, s, maxq


class RefWALayer(torch.nn.Module):
    """reference: WrapperWALayer, auto_round/wrapper.py:568-612 (activation fake-quant shell left around a tuned layer)"""
    # The reference's WrapperWALayer.forward hands the calibrated maximum to the quant function as `act_max=` (wrapper.py:624-633),
    # a keyword nv_fp4_with_static_gs does not have (nvfp.py:102: `tensor_max`), so it lands in **kwargs and every forward AFTER
    # tuning (the quantised-output forward that feeds the next block) uses the batch's own maximum instead.  The product keeps
    # the calibrated static scale there (it is what the checkpoint's input_global_scale tells inference to use);
    # tests that pin the flow against the real reference switch this on.
    follow_reference_ignored_act_max = False

    def __init__(self, layer):
        super().__init__()
        self.orig_layer = layer

    def forward(self, x):
        xq = act_fake_quant(x, self.orig_layer, use_act_max=not RefWALayer.follow_reference_ignored_act_max)
        return F.linear(xq, self.orig_layer.weight, self.orig_layer.bias)


def wrap_block(block, enable_minmax_tuning=True, wrapper_cls=None) -> List[str]:
    """reference: wrapper_block, auto_round/wrapper.py:774-828"""
    wrapper_cls = wrapper_cls or RefWrapperLinear
    names = []
    for n, m in list(block.named_modules()):
        if (isinstance(m, torch.nn.Linear) or _is_conv1d(m)) and int(getattr(m, "bits", 16)) < 16:
            parent = block
            parts = n.split(".")
            for p in parts[:-1]:
                parent = getattr(parent, p)
            setattr(parent, parts[-1], wrapper_cls(m, enable_minmax_tuning))
            names.append(n)
    return names


def unwrap_block(block, best):
    """reference: unwrapper_block, auto_round/wrapper.py:861-878"""
    for n, m in list(block.named_modules()):
        if isinstance(m, RefWrapperLinear):
            parent = block
            parts = n.split(".")
            for p in parts[:-1]:
                parent = getattr(parent, p)
            setattr(parent, parts[-1], m.unwrap(best.get(n) if best else None))


@torch.no_grad()
def sign_sgd_step(params, lr: float, momentum: float = 0.0, state: Optional[dict] = None):
    """reference: SignSGD._single_tensor_sgd, algorithms/quantization/sign_round/sign_sgd.py:356-389 (weight_decay 0, dampening 0,
    no nesterov): with momentum the sign is taken of the running buffer, which starts as a copy of the first gradient."""
    for p in params:
        if p.grad is None:
            continue
        d = p.grad
        if momentum:
            buf = state.get(id(p))
            if buf is None:
                buf = state[id(p)] = d.detach().clone()
            else:
                buf.mul_(momentum).add_(d, alpha=1.0)
            d = buf
        p.add_(torch.sign(d), alpha=-lr)


def linear_lr_stream(lr0: float, iters: int) -> List[float]:
    """The fp32 learning rates LinearLR(1.0 -> 0.0, total_iters=iters) hands to the optimizer, by running the real
    torch scheduler on a dummy parameter (values are pinned by tests/golden/step_*.npz `lr_stream`).
    reference: LinearLR(start_factor=1, end_factor=0, total_iters=iters), sign_round/quantizer.py:419-429"""
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.SGD([{"params": [p], "lr": torch.tensor(lr0)}], lr=lr0)
    sch = torch.optim.lr_scheduler.LinearLR(opt, start_factor=1.0, end_factor=0.0, total_iters=iters)
    out = []
    for _ in range(iters):
        out.append(float(opt.param_groups[0]["lr"]))
        opt.step()
        sch.step()
    return out


class Sampler:
    """Same draws from Python's global `random` as the reference IndexSampler (compressors/utils.py:388-438)."""

    def __init__(self, n, bs):
        self.n, self.bs, self.i = n, bs, 0
        self.idx = list(range(n))
        random.shuffle(self.idx)

    def next_batch(self):
        if self.i + self.bs > self.n:
            random.shuffle(self.idx)
            self.i = 0
        b = self.idx[self.i:self.i + self.bs]
        self.i += self.bs
        return b


def tune_block(block, inputs: torch.Tensor, targets: torch.Tensor, input_others: dict, *, iters=200, batch_size=8,
               lr=None, enable_minmax_tuning=True, amp_dtype=torch.bfloat16, forward=None, record=None,
               max_iters_to_run=None, input_ids=None, amp=True, alg_ext=False, gradient_accumulate_steps=1, minmax_lr=None,
               not_use_best_mse=False, dynamic_max_gap=-1, momentum=0.0):
    """The reference's quantize_block loop in plain torch.  inputs/targets: [N, S, H].  Returns best_params and
    leaves the block unwrapped with baked weights.  `forward(block, x, others)` defaults to block(x, **others)[0].
    reference: SignRoundQuantizer.quantize_block, algorithms/quantization/sign_round/quantizer.py:311-552"""
    use_outlier_loss = False
    wrapper_cls = RefWrapperLinear
    if alg_ext:     # SignRoundV2Quantizer.prepare_run: optimized wrapper for sym int/mx/nv; outlier loss for <4 bits / A4
        first = next((m for m in block.modules() if isinstance(m, torch.nn.Linear) and int(getattr(m, "bits", 16)) < 16), None)
        if first is not None and bool(first.sym):
            wrapper_cls = RefOptWrapperLinear
            use_outlier_loss = int(getattr(first, "act_bits", 16)) <= 4 or int(first.bits) < 4
    names = wrap_block(block, enable_minmax_tuning, wrapper_cls)
    wrappers = {n: m for n, m in block.named_modules() if isinstance(m, RefWrapperLinear)}
    lr0 = lr if lr is not None else 1.0 / iters
    lrs = linear_lr_stream(lr0, iters)
    lrs_mm = lrs if minmax_lr is None else linear_lr_stream(minmax_lr, iters)     # minmax_lr defaults to lr (config.py:110-140)
    params_v = [p for w in wrappers.values() for k, p in w.params.items() if k == "value"]
    params_mm = [p for w in wrappers.values() for k, p in w.params.items() if k != "value"]
    mom_state: dict = {}
    params = params_v + params_mm
    # micro-batches: the sampler draws global batches of batch_size * gradient_accumulate_steps, the loss becomes a SUM that is
    # normalised for reporting only, the gradient accumulates over the micro-batches (quantizer.py:436-452, :470-500)
    accum = gradient_accumulate_steps != 1
    global_bs = min(inputs.shape[0], batch_size * gradient_accumulate_steps)
    sampler = Sampler(inputs.shape[0], global_bs)
    dev_type = inputs.device.type
    best_loss, best, last_best = float(torch.finfo(torch.float32).max), {}, 0
    mse = torch.nn.MSELoss(reduction="sum" if accum else "mean")
    losses = []
    run = iters if max_iters_to_run is None else min(iters, max_iters_to_run)
    vmask = None
    if input_ids is not None:   # valid-token mask: ids == -100 are excluded (quantization/base.py:257-280)
        ids = input_ids if isinstance(input_ids, torch.Tensor) else torch.cat([t.reshape(1, -1) for t in input_ids], 0)
        vm = (ids.reshape(inputs.shape[0], -1) != -100).to(torch.long)
        if not bool(vm.all()):
            vmask = vm.to(inputs.device)
    for i in range(run):
        gidx = sampler.next_batch()
        num_elm = 1
        if vmask is not None:
            num_elm = int(torch.count_nonzero(vmask[gidx]).item())
        elif accum:
            num_elm = global_bs * inputs[0].numel()
        total = 0.0
        for b0 in range(0, len(gidx), batch_size):
            idx = gidx[b0:b0 + batch_size]
            x = inputs[idx]
            ref = targets[idx]
            with torch.autocast(device_type=dev_type, dtype=amp_dtype, enabled=amp):
                out = forward(block, x, input_others) if forward else block(x, **input_others)
                if isinstance(out, (tuple, list)):
                    out = out[0]
            if use_outlier_loss:
                loss = outlier_loss(out, ref.to(out.dtype), None if vmask is None else vmask[idx].unsqueeze(-1))
            elif vmask is not None and not alg_ext:
                # (SignRoundV2Quantizer._get_loss hands the base loss NO valid-token mask when the outlier-suppressed loss is off --
                #  `super()._get_loss(pred, ref, indices, mse_loss, device)`, sign_roundv2/quantizer.py:399 -- so with the algorithm
                #  extension and e.g. an asymmetric scheme every position enters the loss, while num_elm still counts valid tokens)
                m = vmask[idx].unsqueeze(-1)
                loss = mse((out * m).to(torch.float32), (ref * m).to(torch.float32))
            else:
                loss = mse(out.to(torch.float32), ref.to(torch.float32))
            num_elm = 1 if num_elm <= 0 else num_elm
            total += loss.item() / num_elm
            (loss * 1000).backward()
        losses.append(total)
        if total < best_loss:
            best_loss = total
            if not not_use_best_mse:
                last_best = i
                best = {n: {k: p.data.clone() for k, p in w.params.items()} for n, w in wrappers.items()}
        if not_use_best_mse and i == iters - 1:        # the parameters as they stand BEFORE the last step (quantizer.py:513-514)
            best = {n: {k: p.data.clone() for k, p in w.params.items()} for n, w in wrappers.items()}
        if record is not None:
            record(i, wrappers, total)
        if not not_use_best_mse and 0 < dynamic_max_gap <= i - last_best:
            break
        sign_sgd_step(params_v, lrs[i], momentum, mom_state)
        sign_sgd_step(params_mm, lrs_mm[i], momentum, mom_state)
        for p in params:
            p.grad = None
    unwrap_block(block, best)
    return best, dict(losses=losses, best_loss=best_loss, best_iter=last_best, names=names)
