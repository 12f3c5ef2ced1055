"""Registration of the MI355X hot path with the reference's own plugin points, for when
auto-round itself is installed next to this package.  Nothing here is needed for standalone use.

    import auto_round_amd.plugin as p; p.register()
    AutoRound(model, tokenizer, scheme="W4A16", alg_configs=p.MI355XSignRoundConfig(iters=200), device_map=0)
    # or by alias:  AutoRound(..., alg_configs="mi355x_signround")

P1  algorithm registry   register_pipeline_member(ConfigCls) / register_algorithm(name, aliases, config_factory)
                         (auto_round/algorithms/registry.py:56-87,163-168)
P2  wrapper injection    the quantizer attribute `wrapper_block` (sign_round/quantizer.py:66)
P4  export              OutputFormat.register(*names) (export/formats/base.py:119-129): `register_formats()` re-registers the
                         reference's own "auto_round*", "auto_gptq" and "auto_awq" format names with subclasses of the reference's format
                         classes whose `pack_layer` runs the HIP packers (ar_pack_int / ar_pack_awq / ar_pack_fp4) on layers that live on
                         a HIP device -- `quantize_and_save()` then packs on the GPU with no edit of the reference; buffer names,
                         shapes, dtypes and words are the reference packer's (export_to_autoround/export.py:143-239), so
                         `save_quantized` and the inference stack see no difference.  Everything the HIP packers do not cover (Conv2d,
                         W4A8 activation-quant containers, gptqmodel / mlx / fp8 backends, CPU-resident runs) falls through to the
                         reference's own `pack_layer`, i.e. to the code that would have run without this package.
"""
from __future__ import annotations

MI355XSignRoundConfig = None
MI355XSignRoundQuantizer = None


def register():
    """Idempotent.  Raises ImportError when auto_round is not importable.  Uses the reference's own registries:
    `register_pipeline_member` / `register_algorithm` (auto_round/algorithms/registry.py:205-235,312-327)."""
    global MI355XSignRoundConfig, MI355XSignRoundQuantizer
    if MI355XSignRoundQuantizer is not None:
        return MI355XSignRoundConfig, MI355XSignRoundQuantizer

    from auto_round.algorithms.quantization.sign_round.config import SignRoundConfig as _RefConfig
    from auto_round.algorithms.quantization.sign_round.quantizer import SignRoundQuantizer as _RefQuantizer
    from auto_round.algorithms.registry import register_algorithm, register_pipeline_member

    class _Config(_RefConfig):
        """Same fields as the reference SignRoundConfig; selects the MI355X quantizer.  Being a subclass it is not
        coerced to the V2/Adam variants by normalize_algorithm_config (registry.py: `type(config) is SignRoundConfig`).

        Two extra, MI355X-only keys: `fused_block` (None = follow the front door's `enable_torch_compile`, the reference's switch
        for its own compiled block forward; True / False = force the fused HIP block path and the MFMA weight-gradient GEMM on /
        off) and `exact_rounding` (default on: Llama-family blocks through first-party kernels that keep the eager path's bits,
        proven against the module code per kind of block -- auto_round_amd/exact_block.py; ignored when the fused path is asked
        for)."""

        def __init__(self, *, fused_block=None, exact_rounding=True, **kwargs):
            super().__init__(**kwargs)
            self.fused_block = fused_block
            self.exact_rounding = bool(exact_rounding)

    @register_pipeline_member(_Config)
    class _Quantizer(_RefQuantizer):
        """Reference lifecycle (bind / prepare_run / dispatch_block / finalize_run) inherited unchanged; only the
        per-block tuning step is replaced by the HIP path."""

        def register_fp_input_forward_hooks(self, block):
            """With enable_alg_ext the importance matrix is collected while the composer runs the fp forward."""
            handles = super().register_fp_input_forward_hooks(block)
            if getattr(self, "enable_alg_ext", False):
                from .quantizer import SignRoundV2Quantizer

                v2 = SignRoundV2Quantizer.__new__(SignRoundV2Quantizer)
                v2._scheme = SignRoundV2Quantizer._block_scheme(block)
                handles.extend(SignRoundV2Quantizer.register_fp_input_forward_hooks(v2, block))
            return handles

        def quantize_block(self, block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids=None, **kw):
            import torch

            from auto_round.utils.device_manager import device_manager

            from .quantizer import SignRoundConfig, SignRoundQuantizer, SignRoundV2Quantizer

            c = self._config
            fused = getattr(c, "fused_block", None)
            if fused is None:
                fused = bool(getattr(getattr(self, "compress_context", None), "enable_torch_compile", False))
            cfg = SignRoundConfig(
                fused_block=bool(fused), mfma_dw_gemm=bool(fused), exact_rounding=bool(getattr(c, "exact_rounding", True)) and not fused,
                iters=self.iters, lr=None if getattr(c, "lr_is_auto", False) else self.lr,
                minmax_lr=None if getattr(c, "minmax_lr_is_auto", False) else self.minmax_lr,
                lr_scheduler=self.lr_scheduler, momentum=getattr(self, "momentum", 0.0) or 0.0, enable_minmax_tuning=self.enable_minmax_tuning,
                enable_norm_bias_tuning=self.enable_norm_bias_tuning,
                gradient_accumulate_steps=self.gradient_accumulate_steps, not_use_best_mse=self.not_use_best_mse,
                dynamic_max_gap=self.dynamic_max_gap, enable_quanted_input=self.enable_quanted_input,
                batch_size=self.calibration_context.batch_size, bits=getattr(c, "bits", None),
                amp=bool(self.model_context.amp), amp_dtype=self.model_context.amp_dtype or torch.bfloat16)
            # enable_alg_ext=True selects the algorithm extension (searched init scales, imatrix, outlier-suppressed loss)
            q_cls = SignRoundV2Quantizer if getattr(self, "enable_alg_ext", False) else SignRoundQuantizer
            q = self.__dict__.get("_mi355x_engine")      # one engine per run: the exact_rounding proof is made once per kind of block
            if q is None or type(q) is not q_cls or q.device != torch.device(device_manager.device) or q.config != cfg:
                q = self.__dict__["_mi355x_engine"] = q_cls(cfg, device=device_manager.device)
            best = q.quantize_block(block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids=input_ids)
            adopt_act_quant_shells(block, device_manager.device)
            st = q.last_stats
            try:
                from auto_round.logger import logger

                path = "fused block" if q.last_fused_block and not q.last_exact else ("module path" if not q.last_exact else "exact_rounding " + ",".join(
                    f"{k}={v}" if k.startswith("dw_") else k for k, v in sorted((st.get("exact_plan") or {}).items()) if v))
                logger.infoclean(
                    f"quantized {st['quantized']}/{st['quantized'] + st['unquantized']} layers in the block, "
                    f"loss iter 0: {st['init_loss']:.6f} -> iter {st['best_iter']}: {st['best_loss']:.6f} [{path}]")
            except Exception:  # pragma: no cover
                pass
            return best

    register_algorithm("mi355x_signround", aliases=("mi355x", "signround_mi355x"), config_factory=_Config,
                       summary="SignRound block tuning on MI355X (hand-written HIP kernels, auto_round_amd)")
    register_formats()
    _Config.__name__ = "MI355XSignRoundConfig"
    _Quantizer.__name__ = "MI355XSignRoundQuantizer"
    MI355XSignRoundConfig, MI355XSignRoundQuantizer = _Config, _Quantizer
    return _Config, _Quantizer


def adopt_act_quant_shells(block, device="cpu") -> int:
    """After an activation-quantised block (MXFP4 / NVFP4 / W4A8 ...) is tuned, the reference expects ITS `WrapperWALayer` around every
    tuned layer: its exporters unwrap exactly that class (`export_to_nvfp_mx.pack_layer`, export_to_autoround/export_to_nvfp_mx.py:
    60-70) and its quantised-output forward calls `orig_layer.act_quant_func`.  Swap this package's shells for the reference's and give
    the layers the attributes the reference's own unwrapper leaves behind (wrapper.py:432-468).  Returns the number of shells swapped."""
    import torch
    from auto_round.data_type.utils import get_quant_func
    from auto_round.wrapper import WrapperWALayer as _RefShell

    from .wrapper import WrapperWALayer, _set_module

    n_swapped = 0
    for name, m in list(block.named_modules()):
        if not isinstance(m, WrapperWALayer):
            continue
        layer = m.orig_layer
        layer.act_quant_func, layer.act_data_type = get_quant_func(layer.act_data_type, layer.act_bits, layer.act_sym,
                                                                   disable_opt_rtn=True, iters=getattr(layer, "iters", 200))
        sdt = getattr(layer, "scale_dtype", torch.float16)
        layer.q_scale_thresh = 1e-8 if sdt == torch.float32 else 1e-5
        layer.act_min_scale, layer.act_max_scale = torch.tensor(1.0), torch.tensor(1.0)
        _set_module(block, name, _RefShell(layer, enable_torch_compile=False, device=device))
        n_swapped += 1
    return n_swapped


HIP_PACK_STATS = {"hip": 0, "reference": 0}       # layers packed by the HIP packers / handed to the reference's own pack_layer


def _hip_available() -> bool:
    import torch

    return torch.cuda.is_available()


def hip_pack_layer(layer_name: str, model, backend: str, device=None) -> bool:
    """The reference's `pack_layer(layer_name, model, backend, device)` (export/export_to_autoround/export.py:143-239,
    export_to_autogptq/export.py:120-185, export_to_awq/export.py:96-143) with the HIP packers doing the packing.  Same
    post-conditions: the tuned `nn.Linear` at `layer_name` is replaced by a module carrying the packed buffers under the reference's
    names (qweight / qzeros / scales [/ g_idx / bias]; weight_packed / weight_scale ... for fp4), left on the device the layer was on,
    and the original layer's tensors are released.  -> True when handled; False when this layer / backend is not the HIP packers'
    business and the caller should run the reference's own pack_layer (nothing has been touched then)."""
    import torch
    from auto_round.utils import check_to_quantized, get_module, set_module

    from .export import pack_layer as _pack

    if not _hip_available():
        return False
    layer = get_module(model, layer_name)
    if hasattr(layer, "orig_layer"):
        layer = layer.orig_layer
    if type(layer) is not torch.nn.Linear or layer.weight.device.type == "meta":      # packed already, Conv1D / Conv2d, or not materialised
        return False
    dt = str(getattr(layer, "data_type", "int"))
    fp4 = dt.startswith(("mx_fp", "nv_fp")) and int(getattr(layer, "bits", 16)) == 4
    if not fp4 and (int(getattr(layer, "act_bits", 16)) <= 8 or "int" not in dt or dt.startswith("mx")):
        return False                                     # W4A8 containers (pack_qact_layer), mx_int, fp8 ...: the reference's packers
    if any(k in backend for k in ("gptqmodel", "mlx", "fp8", "gguf")):
        return False
    if fp4 and not backend.startswith(("auto_round", "llm_compressor")):
        return False
    if not check_to_quantized(layer):
        return False
    if not (hasattr(layer, "scale") and layer.scale is not None):
        return False
    bits = int(layer.bits)
    if not fp4 and (bits not in (2, 3, 4, 8) or layer.in_features % 32 or layer.out_features % 32
                    or ("awq" in backend and bits != 4)):
        return False
    orig_device = layer.weight.device
    try:        # (a CPU-resident layer -- the orchestrator moves a finished block off the GPU before it packs -- is copied over by the packer)
        qlayer = _pack(layer, backend, device=device if (device is not None and torch.device(device).type == "cuda") else None)
    except (NotImplementedError, ValueError):
        return False
    qlayer.device = orig_device
    qlayer.to(orig_device)
    set_module(model, layer_name, qlayer)
    try:
        from auto_round.export.utils import release_layer_safely

        release_layer_safely(layer)
    except Exception:  # pragma: no cover  (older reference trees keep the helper elsewhere)
        layer.weight = None
    return True


def register_formats():
    """P4 through the reference's own registry (`OutputFormat.register`, export/formats/base.py:119-129): every name the reference
    registered for its AutoRound / AutoGPTQ / AutoAWQ format classes is re-registered with a subclass whose `pack_layer` tries the HIP
    packers first.  Idempotent.  -> {format name: class}"""
    from auto_round.export.formats.backends.auto_awq import AutoAWQFormat
    from auto_round.export.formats.backends.auto_gptq import AutoGPTQFormat
    from auto_round.export.formats.backends.autoround import AutoRoundFormat
    from auto_round.export.formats.base import OutputFormat

    class _HipPackLayer:
        mi355x_hip_packers = True

        def pack_layer(self, layer_name, model, device=None, **kwargs):
            if hip_pack_layer(layer_name, model, self.get_backend_name(), device):
                HIP_PACK_STATS["hip"] += 1
                return None
            HIP_PACK_STATS["reference"] += 1
            return super().pack_layer(layer_name, model, device=device, **kwargs)

    made = {}
    for base in (AutoRoundFormat, AutoGPTQFormat, AutoAWQFormat):
        names = [n for n, c in OutputFormat._format_list.items() if c is base]
        if not names:                                    # registered before: leave it
            continue
        cls = type("MI355X" + base.__name__, (_HipPackLayer, base), {
            "__doc__": f"{base.__name__} with `pack_layer` on the MI355X HIP packers (auto_round_amd.plugin.hip_pack_layer); everything "
                       f"else -- scheme checks, format resolution, save_quantized -- inherited from the reference unchanged."})
        OutputFormat.register(*names)(cls)
        for n in names:
            made[n] = cls
    return made


def packing_quant_linear(backend: str, bits: int, group_size: int, sym: bool):
    """Drop-in for export.dynamic_import_quant_linear_for_packing (export_to_autoround/export.py:56-95)."""
    from .export import dynamic_import_quant_linear_for_packing

    return dynamic_import_quant_linear_for_packing(backend, bits, group_size, sym)
