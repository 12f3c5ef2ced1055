"""Packing side of the hot path: mirrors of the reference's pack-only `QuantLinear` duck type
(`QuantLinear(bits, group_size, infeatures, outfeatures, bias, weight_dtype=...).pack(linear, scales, zeros, g_idx,
device)`; export/export_to_autoround/export.py:206-228) backed by the HIP packers.

  QuantLinearZP   <- auto_round_extension/torch/qlinear_torch_zp.py  ("auto_round:auto_gptq", sym int; stores zp-1)
  QuantLinearPlain<- auto_round_extension/torch/qlinear_torch.py     ("auto_round", asym int; stores zp)
  QuantLinearFP4  <- auto_round/export/export_to_autoround/qlinear_fp.py (MXFP4 / NVFP4 nibbles + e8m0/e4m3 scales)

Buffers have the reference's names, shapes and dtypes (qweight int32 [in/32*bits, out], qzeros int32
[in/gs, out/32*bits], scales fp16 [in/gs, out]; weight_packed uint8 [out, in/2], weight_scale uint8/e4m3) and are
bit-identical to the reference's for the same baked layer (tests/test_gpu_kernels.py::test_pack_*).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import ops


def _pack_device(device, weight: torch.Tensor) -> torch.device:
    """Where a layer is packed: the HIP device asked for, else the one its weight lives on, else the current HIP device (the packers
    are HIP kernels: a CPU-resident layer is moved over for the pack, as the reference's `pack(..., device=...)` does)."""
    dev = torch.device(device) if device is not None else weight.device
    if dev.type != "cuda":
        dev = weight.device if weight.is_cuda else torch.device("cuda", torch.cuda.current_device())
    return dev


class _QuantLinearInt(torch.nn.Module):
    ZP_OFF = 1
    QUANT_TYPE = "mi355x"

    def __init__(self, bits, group_size, infeatures, outfeatures, bias=False, weight_dtype=torch.bfloat16, g_idx=False,
                 **kwargs):
        super().__init__()
        if bits not in (2, 3, 4, 8):
            raise NotImplementedError("Only 2,3,4,8 bits are supported.")
        if infeatures % 32 or outfeatures % 32:
            raise NotImplementedError("in_features and out_features must be divisible by 32.")
        self.bits, self.infeatures, self.outfeatures = bits, infeatures, outfeatures
        self.group_size = group_size if group_size != -1 else infeatures
        ng = (infeatures + self.group_size - 1) // self.group_size
        self.register_buffer("qweight", torch.zeros((infeatures // 32 * bits, outfeatures), dtype=torch.int32))
        self.register_buffer("qzeros", torch.zeros((ng, outfeatures // 32 * bits), dtype=torch.int32))
        self.register_buffer("scales", torch.zeros((ng, outfeatures), dtype=torch.float16))
        if g_idx:      # the "auto_gptq" checkpoint layout carries the (trivial, no act-order) group index of every input channel
            self.register_buffer("g_idx", (torch.arange(infeatures, dtype=torch.int64) // self.group_size).to(torch.int32))
        if bias:
            self.register_buffer("bias", torch.zeros((outfeatures,), dtype=torch.float16))
        else:
            self.bias = None

    def pack(self, linear, scales, zeros, g_idx=None, device=None):
        dev = _pack_device(device, linear.weight)
        W = linear.weight.data.to(dev)
        if W.dim() == 4:
            W = W.flatten(1)
        try:
            from transformers.pytorch_utils import Conv1D

            if isinstance(linear, Conv1D):
                W = W.t()
        except Exception:  # pragma: no cover
            pass
        W = W.contiguous()
        s = scales.to(dev).contiguous()
        z = zeros.to(dev) if isinstance(zeros, torch.Tensor) else zeros
        qw, qz, st = ops.pack_int(W, s, z, gs=self.group_size, bits=self.bits, zp_off=self.ZP_OFF)
        self.qweight, self.qzeros, self.scales = qw, qz, st
        if getattr(linear, "bias", None) is not None:
            self.bias = linear.bias.detach().to(dev).clone().half()


class QuantLinearZP(_QuantLinearInt):
    """GPTQ "zp-1" convention (reference qlinear_torch_zp.py:129,143)."""
    ZP_OFF = 1


class QuantLinearPlain(_QuantLinearInt):
    """stores the zero point unchanged (reference qlinear_torch.py)."""
    ZP_OFF = 0


class WQLinear_GEMM(torch.nn.Module):
    """AWQ GEMM container, the reference's default for W4 asym (export/export_to_awq/utils.py:139-274)."""

    def __init__(self, w_bit, group_size, in_features, out_features, bias, dev=None, training=False):
        super().__init__()
        if w_bit != 4:
            raise NotImplementedError("Only 4-bit are supported for now.")
        self.w_bit, self.in_features, self.out_features = w_bit, in_features, out_features
        self.group_size = group_size if group_size != -1 else in_features
        if in_features % self.group_size or out_features % 8:
            raise ValueError("shape mismatch for the AWQ container")
        self.bias = None

    @classmethod
    def from_linear(cls, linear, w_bit, group_size, init_only=False, scales=None, zeros=None, device=None):
        """scales / zeros arrive TRANSPOSED ([in/gs, out]) like in the reference's pack_layer (export_to_awq/export.py:129-133)."""
        q = cls(w_bit, group_size, linear.in_features, linear.out_features, linear.bias is not None)
        if init_only:
            return q
        if scales is None or zeros is None:
            raise ValueError("Both 'scales' and 'zeros' must be provided (not None)")
        dev = _pack_device(device, linear.weight)
        s2d = scales.to(dev).t().contiguous()
        z = zeros.to(dev).t().contiguous() if isinstance(zeros, torch.Tensor) else zeros
        qw, qz, st = ops.pack_awq(linear.weight.data.to(dev).contiguous(), s2d, z, gs=q.group_size)
        # registered buffers, like the reference's container (export_to_awq/utils.py:166-196): the reference's own save path
        # (`state_dict()` of the packed module) must see them when this class packs behind ITS front door (plugin.register_formats)
        for name, t in (("qweight", qw), ("qzeros", qz), ("scales", st)):
            q.register_buffer(name, t)
        if linear.bias is not None:
            q.__dict__.pop("bias", None)        # (the constructor's placeholder attribute)
            q.register_buffer("bias", linear.bias.detach().to(dev).clone().half())
        return q


class QuantLinearFP4(torch.nn.Module):
    """reference: export/export_to_autoround/qlinear_fp.py:141-265 (`QuantLinear.pack`, `_pack_fp4_to_uint8`): MXFP4 / NVFP4 nibbles
    in `weight_packed`, e8m0 / e4m3 `weight_scale`, NVFP4 `weight_global_scale` / `input_global_scale`."""

    def __init__(self, bits, group_size, infeatures, outfeatures, bias=False, data_type="mx_fp", **kwargs):
        super().__init__()
        if bits != 4:
            raise NotImplementedError("QuantLinearFP4 packs 4-bit MXFP4/NVFP4 only")
        self.is_mx = data_type.startswith("mx")
        self.is_nv = data_type.startswith("nv")
        self.bits, self.group_size, self.infeatures, self.outfeatures = bits, group_size, infeatures, outfeatures
        self.register_buffer("weight_packed", torch.zeros((outfeatures, infeatures // 2), dtype=torch.uint8))
        self.bias = None

    def pack(self, linear, scales, zeros=None, g_idx=None, global_scale=None, input_global_scale=None, device=None):
        dev = _pack_device(device, linear.weight)
        W = linear.weight.data.to(dev).contiguous()
        mode = 0 if self.is_mx else 1
        s = scales.to(dev).contiguous()
        gsc = None
        if self.is_nv:
            gsc = global_scale.to(device=dev, dtype=torch.float32).reshape(1)
            s = s.to(torch.float32)
        packed, sb = ops.pack_fp4(W, s.reshape(-1), mode=mode, gs=self.group_size, global_scale=gsc)
        self.weight_packed = packed
        # registered buffers (the reference's qlinear_fp.QuantLinear registers them in __init__): `state_dict()` is what its save path reads
        self._set_buffer("weight_scale", sb if self.is_mx else sb.view(torch.float8_e4m3fn))
        if gsc is not None:
            self._set_buffer("weight_global_scale", gsc)
        if input_global_scale is not None:
            self._set_buffer("input_global_scale", input_global_scale.to(torch.float32).to(dev).reshape([1]))
        if getattr(linear, "bias", None) is not None:
            self._set_buffer("bias", linear.bias.detach().to(dev).to(torch.float16))

    def _set_buffer(self, name, t):
        if name in self._buffers:
            self._buffers[name] = t
        else:
            if name in self.__dict__:
                del self.__dict__[name]
            self.register_buffer(name, t)


def _is_conv1d(m) -> bool:
    try:
        from transformers.pytorch_utils import Conv1D
    except Exception:  # pragma: no cover
        return False
    return isinstance(m, Conv1D)


def dynamic_import_quant_linear_for_packing(backend: str, bits: int, group_size: int, sym: bool, act_bits: int = 16):
    """reference: export/export_to_autoround/export.py:56-95 -- which packer a backend string selects."""
    if "auto_round" in backend and "awq" not in backend and "gptq" not in backend:
        return QuantLinearPlain
    if "gptq" in backend and "gptqmodel" not in backend:     # "auto_round:auto_gptq" and the plain "auto_gptq" format
        return QuantLinearZP                                  # (export/utils.py:314-331 picks the same zp-1 packer for both)
    raise ValueError(f"unsupported backend for the MI355X packers: {backend}")


def pack_layer(layer: torch.nn.Linear, backend: str = "auto_round:auto_gptq", device=None):
    """reference: export/export_to_autoround/export.py:143-239 for one already-unwrapped layer carrying
    `scale` / `zp`.  Returns the packed QuantLinear (does not splice it into a model)."""
    bits, gs, sym = int(layer.bits), int(layer.group_size), bool(layer.sym)
    conv1d = _is_conv1d(layer)
    # transformers' Conv1D (GPT-2) stores its weight [in, out]; the packers work on [out, in] like the reference's exporters
    # (export_to_autoround/export.py:200-205 takes in/out features from the transposed shape, QuantLinear.pack transposes)
    in_f, out_f = (layer.weight.shape if conv1d else layer.weight.shape[::-1])
    dt = str(getattr(layer, "data_type", "int"))
    if conv1d and (dt.startswith(("mx_fp", "nv_fp")) or "awq" in backend):
        raise NotImplementedError("Conv1D layers are packed in the auto_round / auto_gptq INT layouts only")
    if dt.startswith("mx_fp") or dt.startswith("nv_fp"):      # export_to_nvfp_mx.pack_layer -> qlinear_fp.QuantLinear.pack
        # the llm_compressor ("compressed-tensors" nvfp4/mxfp4-pack-quantized) exporter packs through the same
        # QuantLinear and only adds the static input scale (export_to_llmcompressor/export_to_fp.py:68-131)
        input_gs = getattr(layer, "input_global_scale", None)
        adt = str(getattr(layer, "act_data_type", ""))
        if adt.startswith("nv_fp") and int(getattr(layer, "act_bits", 16)) <= 8 and input_gs is None \
                and getattr(layer, "act_max", None) is not None:
            amax = torch.as_tensor(layer.act_max, dtype=torch.float32).abs().max()
            input_gs = torch.where(amax == 0, torch.zeros_like(amax), (448.0 * 6.0) * (1.0 / amax))   # calculate_gparam
            layer.input_global_scale = input_gs
            del layer.act_max
        ql = QuantLinearFP4(bits, gs, in_f, out_f, bias=layer.bias is not None, data_type=dt)
        ql.pack(layer, layer.scale, global_scale=getattr(layer, "weight_global_scale", None),
                input_global_scale=input_gs, device=device)
        return ql
    if "awq" in backend:      # export_to_awq.pack_layer (export.py:114-143)
        scale, zp = layer.scale.t().contiguous(), layer.zp
        if isinstance(zp, torch.Tensor):
            zp = zp.t().contiguous().to(torch.float32)
            if sym:
                zp = int(zp.flatten()[0])
        return WQLinear_GEMM.from_linear(layer, bits, gs, scales=scale, zeros=zp, device=device)
    QL = dynamic_import_quant_linear_for_packing(backend, bits, gs, sym)
    ql = QL(bits, gs, in_f, out_f, bias=layer.bias is not None, weight_dtype=layer.weight.dtype,
            g_idx=not backend.startswith("auto_round"))       # export_to_autogptq/export.py:163
    zp = layer.zp
    if sym and isinstance(zp, torch.Tensor) and QL is QuantLinearPlain:
        zp = int(zp.flatten()[0])
    ql.pack(layer, layer.scale, zp, None, device=device)
    return ql


def pack_block(block, backend: Optional[str] = None) -> Dict[str, torch.nn.Module]:
    """Pack every tuned linear of an unwrapped block (the orchestrator's immediate_pack, orchestrator.py:327-337).
    backend defaults to what AutoRoundFormat picks: sym int -> auto_round:auto_gptq, W4 asym -> auto_round:auto_awq,
    other asym -> auto_round (export/formats/backends/autoround.py:59-70)."""
    out = {}
    for n, m in block.named_modules():
        if (isinstance(m, torch.nn.Linear) or _is_conv1d(m)) and hasattr(m, "scale") and int(getattr(m, "bits", 16)) < 16:
            be = backend or ("auto_round:auto_gptq" if m.sym else ("auto_round:auto_awq" if int(m.bits) == 4 else "auto_round"))
            out[n] = pack_layer(m, be)
    return out
