"""Standalone caller of the hot path with the reference's front-door names -- `AutoRound(model, tokenizer, scheme=...,
iters=..., nsamples=..., seqlen=..., batch_size=...).quantize() / .save_quantized() / .quantize_and_save()`
(auto_round/autoround.py, compressors/base.py) -- reduced to what a decoder-only HF language model needs:

  1. scheme -> per-layer attributes                         (schemes.py; reference: apply_plan_to_model)
  2. capture the first block's inputs on the calibration tokens (reference: calibration/llm.py cache_inter_data: a forward
     hook on block 0 records hidden_states + the shared kwargs and stops the forward)
  3. tune the blocks in order on the GPU                    (model_tuner.tune_blocks -> SignRound[V2]Quantizer)
  4. pack + stream to safetensors shards                    (export.pack_block, shard_writer.ShardWriter), config.json with
     the reference's `quantization_config` keys for format "auto_round"

Multi-GPU (round 6): launched as one process per GPU (`torchrun --nproc-per-node N ...`, or any launcher that sets RANK / LOCAL_RANK /
WORLD_SIZE), `device_map="0,1,...,N-1"` names the devices and every rank takes the one at its LOCAL_RANK.  With
`enable_quanted_input=False` the blocks are SHARDED over the ranks (sharding.tune_sharded: RCCL broadcast of the calibration
activations, pipelined point-to-point relay of the fp chain, no per-iteration collective; every rank packs and writes the blocks it
tuned into its own shard files, rank 0 writes the rest of the model and the one index); with the default quantised-input chaining the
blocks are sequential and every block is tuned DATA-PARALLEL instead (SignRoundConfig.data_parallel: each rank a share of every
minibatch, one bf16 all-reduce of the weight-gradient buffer per iteration).  The reference's own `device_map="0,1,..."`
(algorithms/quantization/sign_round/quantizer.py:82-118, utils/device.py:923-1040) spreads ONE block over the devices of one process to
make it fit; on 288 GB parts a block always fits, so the same argument buys throughput here.

Not rebuilt (use the reference with `auto_round_amd.plugin` for these): dataset download/tokenisation (no network here:
`dataset` must be token ids), multimodal / diffusion models, AutoScheme, GGUF / FP8 formats, lm_head / embedding quantisation,
low-memory offloading."""
from __future__ import annotations

import json
import os
from typing import Dict, List, Optional, Union

import torch

from .model_tuner import tune_blocks, tune_blocks_sharded
from .quantizer import SignRoundConfig, SignRoundQuantizer, SignRoundV2Quantizer
from .schemes import SCHEME_KEYS, apply_scheme, expand_layer_config, is_quantizable, layer_pattern_regex, resolve_scheme
from .shard_writer import ShardWriter


class _StopForward(Exception):
    pass


def get_block_names(model) -> List[List[str]]:
    """Names of the repeated decoder blocks: the children of every nn.ModuleList found first on each path from the root
    (reference: utils/model.py get_block_names, LLM branch).  The decoder stack is the longest such list."""
    groups = []

    def search(prefix, module):
        for n, m in module.named_children():
            full = f"{prefix}.{n}" if prefix else n
            if isinstance(m, torch.nn.ModuleList):
                groups.append([f"{full}.{i}" for i, _ in m.named_children()])
            else:
                search(full, m)

    search("", model)
    return groups


class AutoRound:
    """The reference's entry object for this path: `AutoRound(model, tokenizer, scheme=..., iters=..., ...)` with `quantize()`,
    `save_quantized()`, `quantize_and_save()` (auto_round/autoround.py:732-793, compressors/base.py quantize / save_quantized /
    quantize_and_save:1926-1990); arguments keep the reference's names, meaning and defaults."""

    def __init__(self, model, tokenizer=None, scheme: Union[str, dict] = "W4A16", *, bits=None, group_size=None, sym=None,
                 act_bits=None, act_group_size=None, act_sym=None, act_dynamic=None, act_data_type=None,
                 iters: int = 200, lr=None, minmax_lr=None, nsamples: int = 128, seqlen: int = 2048, batch_size: int = 8,
                 dataset=None, enable_alg_ext: bool = False, enable_quanted_input: bool = True,
                 enable_minmax_tuning: bool = True, gradient_accumulate_steps: int = 1, not_use_best_mse: bool = False,
                 dynamic_max_gap: int = -1, layer_config: Optional[Dict[str, dict]] = None, device_map=0, seed: int = 42,
                 amp: bool = True, momentum: float = 0.0, **kwargs):
        # the reference's memory knobs change how it runs, not what it computes: accepted and ignored here (everything of one block
        # is resident in HBM).  enable_torch_compile (compressors/base.py:1177-1179: compiled block_forward, "about 20 %") selects its
        # MI355X counterpart: the fused HIP block path and the hand-written MFMA weight-gradient GEMM -- same trade as the
        # reference's: faster, different rounding points inside the block.
        for k in ("low_gpu_mem_usage", "low_cpu_mem_usage"):
            kwargs.pop(k, None)
        # The reference's constructor turns torch's deterministic-algorithms mode ON (compressors/base.py:339-351: warn-only by default,
        # strict with enable_deterministic_algorithms=True).  That is process-global state and part of what the reference COMPUTES: under
        # it torch takes its deterministic forms of index_add_ / index_put_(accumulate) / scatter, and on this stack the library picks
        # other kernels where the default ones accumulate with atomics -- OPT-125M's head-size-64 attention backward, the per-expert
        # GEMMs of a Mixtral block over ragged row counts.  Round 5 found it the hard way: with the mode on (the reference itself, and
        # this package behind the reference's front door) both blocks reproduce the reference bit for bit, without it the same code
        # parts from it after ~55 iterations.  Mirrored here, same keywords, same default.
        # NOTE the side effect: like the reference's, this constructor leaves the mode ON for the whole process -- outside
        # `quantize_block` every `torch.empty` of the caller's process is then NaN-filled (torch.utils.deterministic.
        # fill_uninitialized_memory), calibration and export included.  `keep_torch_determinism_mode=True` (MI355X-only keyword) leaves
        # the process's mode untouched for callers who manage it themselves; results are then only the reference's if they set it too.
        strict = bool(kwargs.pop("enable_deterministic_algorithms", False)) or not bool(kwargs.pop("disable_deterministic_algorithms", True))
        if not bool(kwargs.pop("keep_torch_determinism_mode", False)):
            os.environ.setdefault("CUBLAS_WORKSPACE_CONFIG", ":4096:8")
            torch.use_deterministic_algorithms(True, warn_only=not strict)
        fused = bool(kwargs.pop("enable_torch_compile", False))
        # MI355X-only: Llama-family blocks through the first-party kernels that keep the eager path's bits (exact_block.py) -- on by
        # default: same results as the module code (proven per kind of block before use, module path otherwise), fewer launches
        exact = bool(kwargs.pop("exact_rounding", True)) and not fused
        # The reference's calibrator drives the model with an attention mask whenever the dataset is not one of its named ones
        # (calibration/llm.py:291-293, 362-402: ones, trailing repeats of a sample's last token cleared, the last position ALWAYS
        # cleared so that no model folds an all-ones mask into None) -- a token tensor, the only kind of dataset this front door takes,
        # is that case -- and its input cache casts the boolean [1, 1, S, S] mask transformers builds from it to the activation dtype
        # (calibration/inputs.py:100-107): a 0 / 1 ADDITIVE bias from then on.  That mask is part of what the reference computes (the
        # block's attention is then NOT causal), so it is mirrored by default; `calibration_attention_mask=False` (MI355X-only keyword)
        # calibrates and tunes with plain causal attention instead.
        self.calibration_attention_mask = bool(kwargs.pop("calibration_attention_mask", True))
        # ... and it captures the first block's inputs with the model ON THE CPU (calibration/llm.py:74-90: only the embedding part runs,
        # "also fast on CPU"), so the rotary tables are the host libm's: mirrored by default (`pre_block_modules_on_cpu`);
        # `capture_on_cpu=False` (MI355X-only keyword) keeps that forward on the GPU.
        self.capture_on_cpu = bool(kwargs.pop("capture_on_cpu", True))
        if kwargs.pop("platform", "hf") != "hf":
            raise NotImplementedError("only Hugging Face models (platform='hf') are handled")
        legacy_device = kwargs.pop("device", None)           # autoround.py:753-757: deprecated alias of device_map
        if legacy_device is not None and device_map in (None, 0):
            device_map = legacy_device
        alg = kwargs.pop("algorithm", None) or kwargs.pop("alg_configs", None)
        if alg is not None and str(alg).lower().replace("_", "") not in ("signround", "autoround"):
            raise NotImplementedError(f"algorithm {alg!r}: this path implements SignRound (and its extension, enable_alg_ext=True)")
        if kwargs:
            raise TypeError(f"arguments outside the MI355X hot path: {sorted(kwargs)} (use the reference with auto_round_amd.plugin)")
        self.model, self.tokenizer = model, tokenizer
        self.scheme = resolve_scheme(scheme, bits=bits, group_size=group_size, sym=sym, act_bits=act_bits,
                                     act_group_size=act_group_size, act_sym=act_sym, act_dynamic=act_dynamic,
                                     act_data_type=act_data_type)
        if (self.scheme.get("act_bits") or 16) <= 8:         # the reference resolves unset activation fields from the weights'
            for k, v in (("act_data_type", self.scheme["data_type"]), ("act_sym", self.scheme["sym"]), ("act_dynamic", True),
                         ("act_group_size", self.scheme["group_size"])):
                if self.scheme.get(k) is None:
                    self.scheme[k] = v
        self.nsamples, self.seqlen, self.seed = nsamples, seqlen, seed
        self.dataset = dataset
        self.rank, self.world, local_rank = dist_env()
        self.devices = parse_device_map(device_map)
        if not torch.cuda.is_available() or any(d.type != "cuda" for d in self.devices):
            raise RuntimeError(f"AutoRound (MI355X path) needs a HIP device, got device_map={device_map!r} with "
                               f"torch.cuda.is_available()={torch.cuda.is_available()}; there is no CPU fallback")
        self.device = pick_rank_device(self.devices, local_rank, self.world, torch.cuda.device_count())
        if self.world == 1 and len(self.devices) > 1:
            import warnings

            warnings.warn(f"device_map={device_map!r} names {len(self.devices)} devices but this is a single process: the MI355X path runs one "
                          f"process per GPU (torchrun --nproc-per-node {len(self.devices)} ...); tuning on {self.device} alone")
        # one process per GPU: blocks shard over the ranks when they are independent (fp chain), else every block is tuned data-parallel
        self.sharded = self.world > 1 and not enable_quanted_input
        self.data_parallel = self.world > 1 and bool(enable_quanted_input)
        self.enable_alg_ext = enable_alg_ext
        amp_dtype = next(model.parameters()).dtype
        if amp_dtype not in (torch.bfloat16, torch.float16):
            amp, amp_dtype = False, torch.bfloat16
        self.config = SignRoundConfig(iters=iters, lr=lr, minmax_lr=minmax_lr, batch_size=batch_size, bits=self.scheme["bits"],
                                      enable_minmax_tuning=enable_minmax_tuning, enable_quanted_input=enable_quanted_input,
                                      gradient_accumulate_steps=gradient_accumulate_steps, not_use_best_mse=not_use_best_mse,
                                      dynamic_max_gap=dynamic_max_gap, amp=amp, amp_dtype=amp_dtype, momentum=momentum, fused_block=fused,
                                      mfma_dw_gemm=fused, exact_rounding=exact, data_parallel=self.data_parallel)
        self.layer_config_in = layer_config
        self.layer_config: Dict[str, dict] = {}
        self.block_names: List[str] = []
        self.records: List[dict] = []
        self.quantized = False

    # -- calibration -----------------------------------------------------------------------------------------------------
    def _calibration_tokens(self) -> torch.Tensor:
        ds = self.dataset
        if ds is None or isinstance(ds, str):
            raise ValueError("dataset must be token ids (a [nsamples, seqlen] LongTensor, a list of 1-D/2-D LongTensors or an "
                             "iterable of such batches): there is no network to fetch a named dataset from")
        if isinstance(ds, torch.Tensor):
            rows = [r for r in ds.reshape(-1, ds.shape[-1])]
        else:
            rows = []
            for item in ds:
                t = item["input_ids"] if isinstance(item, dict) else item
                t = torch.as_tensor(t)
                rows.extend(r for r in t.reshape(-1, t.shape[-1]))
        rows = [r[:self.seqlen] for r in rows if r.numel() >= self.seqlen]      # shorter samples are skipped (llm.py:338)
        if len(rows) < 1:
            raise ValueError(f"no calibration sample reaches seqlen={self.seqlen}")
        return torch.stack(rows[:self.nsamples]).long()

    @torch.no_grad()
    def _capture_block0_inputs(self, blocks, tokens):
        first = blocks[0]
        captured, shared = [], {}

        def hook(module, args, kwargs):
            hs = args[0] if args else kwargs["hidden_states"]
            captured.append(hs.detach())
            if not shared:
                bs = hs.shape[0]
                for k, v in kwargs.items():
                    if k in ("hidden_states", "past_key_values", "past_key_value", "use_cache", "cache_position"):
                        continue
                    shared[k] = _first_sample(v, bs)
            raise _StopForward

        masks = []

        def mask_hook(module, args, kwargs):          # (before `hook`: it raises) every batch's own mask rows, as the reference caches them
            m = kwargs.get("attention_mask")
            if isinstance(m, torch.Tensor) and m.dim() == 4:
                masks.append(m.detach())

        hm = first.register_forward_pre_hook(mask_hook, with_kwargs=True) if self.calibration_attention_mask else None
        h = first.register_forward_pre_hook(hook, with_kwargs=True)
        bs = self.config.batch_size
        # the part of the model in front of the blocks runs where the reference runs it: on the CPU (pre_block_modules_on_cpu)
        cap_dev = torch.device("cpu") if self.capture_on_cpu else self.device
        import contextlib

        try:
            with (pre_block_modules_on_cpu(self.model, blocks) if self.capture_on_cpu else contextlib.nullcontext()):
                for b0 in range(0, tokens.shape[0], bs):
                    ids = tokens[b0:b0 + bs].to(cap_dev)
                    kw = {"use_cache": False}
                    if self.calibration_attention_mask:
                        kw["attention_mask"] = calibration_attention_mask(ids)
                    try:
                        self.model(input_ids=ids, **kw)
                    except _StopForward:
                        pass
        finally:
            h.remove()
            if hm is not None:
                hm.remove()
        captured[:] = [c.to(self.device) for c in captured]
        masks[:] = [c.to(self.device) for c in masks]
        shared.update({k: _to_device(v, self.device) for k, v in shared.items()})
        if self.calibration_attention_mask and isinstance(shared.get("attention_mask"), torch.Tensor):
            # the input cache's cast (inputs.py:100-107): boolean / half-precision mask -> the activation dtype, i.e. a 0 / 1 bias
            rows = torch.cat(masks, dim=0) if masks else shared["attention_mask"]
            if rows.shape[0] > 1 and not bool((rows == rows[:1]).all()):
                import warnings

                warnings.warn("calibration samples end in repeated tokens of different lengths: the reference would cache one attention mask "
                              "per sample; this front door keeps ONE shared mask (the last position cleared) for all of them")
                ids = tokens[:1].to(cap_dev)
                am = torch.ones_like(ids)
                am[:, -1] = 0
                try:
                    captured_n = len(captured)
                    masks.clear()
                    hm = first.register_forward_pre_hook(mask_hook, with_kwargs=True)
                    h = first.register_forward_pre_hook(hook, with_kwargs=True)
                    try:
                        with (pre_block_modules_on_cpu(self.model, blocks) if self.capture_on_cpu else contextlib.nullcontext()):
                            self.model(input_ids=ids, attention_mask=am, use_cache=False)
                    except _StopForward:
                        pass
                finally:
                    h.remove()
                    hm.remove()
                    del captured[captured_n:]
                rows = masks[0].to(self.device) if masks else rows
            m = rows[:1]
            amp_dtype = self.config.amp_dtype
            shared["attention_mask"] = m.to(amp_dtype) if (m.dtype == torch.bool or m.is_floating_point()) else m
        return torch.cat(captured, dim=0), shared

    # -- the run ---------------------------------------------------------------------------------------------------------
    def quantize(self):
        """-> (model, layer_config).  The blocks' tuned linears hold the fake-quant weights and carry `scale` / `zp`."""
        import transformers

        transformers.set_seed(self.seed)        # seeds `random` (IndexSampler) like the reference's compressor does
        if self.world > 1:
            init_process_group(self.device)
        model = self.model.to(self.device).eval()
        for p in model.parameters():
            p.requires_grad_(False)
        from .moe_unfuse import unfuse_moe_experts

        self.unfused_moe = unfuse_moe_experts(model)      # fused 3-D expert parameters -> per-expert nn.Linear (moe_unfuse.py)
        groups = get_block_names(model)
        if not groups:
            raise ValueError("no repeated decoder blocks (nn.ModuleList) found in the model")
        self.block_names = max(groups, key=len)
        blocks = [model.get_submodule(n) for n in self.block_names]
        for n, b in zip(self.block_names, blocks):
            for ln, cfg in apply_scheme(b, self.scheme, layer_config=_block_layer_config(self.layer_config_in, n, b)).items():
                self.layer_config[f"{n}.{ln}"] = cfg
        tokens = self._calibration_tokens()
        ids_for_mask = loss_mask_ids(tokens, getattr(self.tokenizer, "pad_token_id", None))
        # grouped-query "sdpa" attention would silently run on the 2x slower flash kernels (attention.py) -- on the fused / plain module
        # paths; with `exact_rounding` the model keeps transformers' own "sdpa" function: that call is what the reference makes, what the
        # exact blocks are proven against (they refuse any other attention function) and what `exact_attention` swaps per block
        cfg_obj, old_attn = getattr(model, "config", None), None
        if (self.config.sdpa_backend == "efficient" and not self.config.exact_rounding
                and getattr(cfg_obj, "_attn_implementation", None) == "sdpa"):
            from .attention import register_mi355x_sdpa

            old_attn, cfg_obj._attn_implementation = cfg_obj._attn_implementation, register_mi355x_sdpa()
        try:
            x0, others = self._capture_block0_inputs(blocks, tokens)
            q_cls = SignRoundV2Quantizer if self.enable_alg_ext else SignRoundQuantizer
            self.quantizer = q_cls(self.config, device=self.device)
            if self.sharded:        # every rank tunes the blocks it owns against the fp chain; then every rank gets every tuned block
                with torch.cuda.device(self.device):
                    recs = tune_blocks_sharded(blocks, x0, others, self.quantizer, seed=self.seed, input_ids=ids_for_mask,
                                               block_names=self.block_names)
                    self.owned_blocks = sorted(recs)
                    self.records = [recs.get(k) for k in range(len(blocks))]
                    sync_tuned_blocks(blocks, self.world, self.device)
            else:
                self.owned_blocks = list(range(len(blocks)))
                self.records = tune_blocks(blocks, x0, others, self.quantizer, input_ids=ids_for_mask,
                                           block_names=self.block_names)
        finally:
            if old_attn is not None:
                cfg_obj._attn_implementation = old_attn
        self.quantized = True
        return model, self.layer_config

    def _quantization_config(self, backend: str) -> dict:
        """Same keys as the reference writes for format "auto_round" (export_to_autoround/export.py:286-330 after
        filter_quantization_config): scheme fields that differ from the 16-bit defaults, iters, block_name_to_quantize,
        packing_format, quant_method, and per-layer deviations in extra_config."""
        qc = {k: self.scheme.get(k) for k in ("bits", "group_size", "sym", "data_type")}
        if (self.scheme.get("act_bits") or 16) <= 8:
            qc.update({k: self.scheme.get(k) for k in ("act_bits", "act_data_type", "act_group_size", "act_sym", "act_dynamic")
                       if self.scheme.get(k) is not None})
        qc.update(quant_method="auto-round", packing_format=backend, iters=self.config.iters,
                  block_name_to_quantize=os.path.commonprefix(self.block_names).rstrip("."), autoround_version="mi355x-0.1.0",
                  static_kv_granularity="tensor", static_attention_granularity="tensor")
        extra = {n: {k: v for k, v in c.items() if c.get(k) != self.scheme.get(k) and v is not None}
                 for n, c in self.layer_config.items() if any(c.get(k) != self.scheme.get(k) for k in SCHEME_KEYS)}
        for pat, over in self._pattern_config().items():         # patterns are kept as such for loaders that re-apply them
            extra[layer_pattern_regex(pat)] = {k: v for k, v in over.items() if v != self.scheme.get(k)}
        if extra:
            qc["extra_config"] = extra
        return qc

    def _pattern_config(self) -> Dict[str, dict]:
        """The user's `layer_config` entries that are patterns rather than layer names (the reference's `regex_config`,
        compressors/layer_config/resolver.py:289-322)."""
        return {k: dict(v) for k, v in (getattr(self, "layer_config_in", None) or {}).items() if k not in self.layer_config}

    def _deviating_layers(self) -> Dict[str, dict]:
        """Layers whose scheme differs from the model-wide one (export_to_autoround/utils.py:21-41 `check_neq_config`)."""
        return {n: c for n, c in self.layer_config.items()
                if any(c.get(k) not in (self.scheme.get(k), None) for k in SCHEME_KEYS)}

    def _gptq_quantization_config(self) -> dict:
        """format "auto_gptq" (export_to_autogptq/export.py:188-325): GPTQ's own keys, per-layer deviations as its
        `dynamic` rules ("+:regex" = quantise with these settings, "-:regex" = leave in 16 bit)."""
        import re

        qc = {k: self.scheme.get(k) for k in ("bits", "group_size", "sym", "data_type")}
        qc.update(iters=self.config.iters, autoround_version="mi355x-0.1.0", static_kv_granularity="tensor",
                  static_attention_granularity="tensor", lm_head=False, provider="auto-round", quant_method="gptq",
                  desc_act=False, true_sequential=False, damp_percent=0.01)
        dynamic, patterns = {}, self._pattern_config()
        rules = [(layer_pattern_regex(p), {**self.scheme, **over}) for p, over in patterns.items()]       # patterns first,
        rules += [(f".*{re.escape(n)}.*", c) for n, c in self._deviating_layers().items()                 # then single layers
                  if not any(re.search(p, n) for p in patterns)]                                          # they do not cover
        for rx, c in rules:
            if int(c.get("bits", 16)) < 16:
                dynamic[f"+:{rx}"] = {k: c[k] for k in ("bits", "group_size", "sym")}
            else:
                dynamic[f"-:{rx}"] = {}
        if dynamic:
            qc["dynamic"] = dynamic
        quantised, skipped = set(), set()
        for bn in self.block_names:
            for n, c in self.layer_config.items():
                if n.startswith(bn + "."):                         # name inside its block
                    (quantised if int(c.get("bits", 16)) <= 8 else skipped).add(n[len(bn) + 1:])
        if skipped:                                                # export.py:267-288: only written for partially tuned blocks
            qc["modules_in_block_to_quantize"] = [sorted(quantised)]
        return qc

    def _llmc_quantization_config(self) -> dict:
        """format "llm_compressor" for MXFP4 / NVFP4 (export_to_llmcompressor/export_to_fp.py:140-156 `_get_scheme` /
        `_get_group_format`, :290-380; config.py:46-101): the `compressed-tensors` QuantizationConfig the reference obtains from
        that package's preset schemes (`preset_name_to_scheme("NVFP4" | "MXFP4", ["Linear"])`, status COMPRESSED) and dumps with
        `to_dict()`, plus `format`, `provider` and the `ignore` list of `generate_ignore_regex_list` (utils.py:21-51) + lm_head.
        compressed-tensors (a third-party dependency of the reference, not vendored and not installed here) cannot be imported
        to produce the dict, so it is restated in the layout the reference itself hard-codes for its NVFP4-E5M3 variant
        (config.py:103-139): NVFP4 = 4-bit float, tensor_group strategy, group 16, static weights / locally dynamic inputs;
        MXFP4 = 4-bit float, group strategy, group 32, static weights / dynamic inputs."""
        dt, bits = str(self.scheme["data_type"]), int(self.scheme["bits"])
        if bits != 4 or not dt.startswith(("mx_fp", "nv_fp")):
            raise ValueError(f"llm_compressor format: implemented for the MXFP4 / NVFP4 schemes, got data_type={dt} bits={bits}")
        nv = dt.startswith("nv_fp")
        mixed = {(int(c.get("bits", 16)), str(c.get("data_type"))) for c in self.layer_config.values() if int(c.get("bits", 16)) <= 8}
        if len(mixed) > 1:
            raise NotImplementedError("llm_compressor format: mixed-precision config groups are not written by the MI355X path")

        def quant_args(dynamic):
            return {"actorder": None, "block_structure": None, "dynamic": dynamic, "group_size": 16 if nv else 32, "num_bits": 4,
                    "observer": "minmax", "observer_kwargs": {}, "strategy": "tensor_group" if nv else "group", "symmetric": True,
                    "type": "float"}

        act = (self.scheme.get("act_bits") or 16) <= 8
        ignore = ["re:" + layer_pattern_regex(p) for p, over in self._pattern_config().items() if int(over.get("bits", 0)) > 8]
        ignore += [n for n, c in self.layer_config.items() if int(c.get("bits", 16)) > 8]
        for n, m in self.model.named_modules():              # get_lm_head_name: the output projection stays out
            if n.split(".")[-1] == "lm_head" and n not in self.layer_config and n not in ignore:
                ignore.append(n)
        fmt = "nvfp4-pack-quantized" if nv else "mxfp4-pack-quantized"
        live = _compressed_tensors_config("NVFP4" if nv else "MXFP4", ignore, fmt) if act else None
        if live is not None:      # the package the reference itself asks: whatever fields its installed version emits
            return live
        return {"config_groups": {"group_0": {"input_activations": quant_args("local" if nv else True) if act else None,
                                              "output_activations": None, "targets": ["Linear"], "weights": quant_args(False)}},
                "format": fmt, "global_compression_ratio": None,
                "ignore": ignore, "kv_cache_scheme": None, "quant_method": "compressed-tensors",
                "quantization_status": "compressed", "provider": "auto-round"}

    def _awq_quantization_config(self) -> dict:
        """format "auto_awq" (export_to_awq/export.py:201-226): AutoAWQ's GEMM keys; layers left in 16 bit are listed."""
        qc = {k: self.scheme.get(k) for k in ("bits", "group_size", "sym", "data_type")}
        keep = [n for n, m in self.model.named_modules()            # linears outside the tuned blocks (lm_head, projectors)
                if is_quantizable(m) and n not in self.layer_config
                and not any(n.startswith(b + ".") for b in self.block_names)]
        keep += [n for n, c in self.layer_config.items() if int(c.get("bits", 16)) > 8]
        keep += [p for p, over in self._pattern_config().items() if int(over.get("bits", 0)) > 8]
        qc.update(iters=self.config.iters, autoround_version="mi355x-0.1.0", static_kv_granularity="tensor",
                  static_attention_granularity="tensor", provider="auto-round", quant_method="awq",
                  to_quant_block_names=os.path.commonprefix(self.block_names).rstrip("."),
                  zero_point=not bool(self.scheme["sym"]), version="gemm", modules_to_not_convert=keep)
        return qc

    @torch.no_grad()
    def save_quantized(self, output_dir: str, format: str = "auto_round", inplace: bool = True,
                       max_shard_bytes: int = 5 * 1024 ** 3, **kwargs):
        """Pack every tuned layer and write the checkpoint: safetensors shards (+ index), config.json with
        `quantization_config`, tokenizer files when a tokenizer was given.  Formats: "auto_round" (default; optionally with
        an explicit ":auto_gptq" / ":auto_awq" packing backend), and for INT schemes the plain "auto_gptq" and "auto_awq"
        layouts (export/formats/backends/auto_gptq.py, auto_awq.py)."""
        # `inplace` (compressors/base.py save_quantized) only says whether the reference may splice its packed modules into
        # the caller's model; the tensors written are the same, and this writer never alters the model
        if kwargs.pop("safe_serialization", True) is not True:
            raise NotImplementedError("checkpoints are written as safetensors")
        if kwargs:
            raise TypeError(f"save_quantized: unsupported arguments {sorted(kwargs)}")
        if format not in ("auto_round", "auto_round:auto_gptq", "auto_round:auto_awq", "auto_gptq", "auto_awq", "llm_compressor"):
            raise NotImplementedError(f"format {format!r}: the MI355X path writes auto_round, auto_gptq, auto_awq and (MXFP4 / NVFP4) "
                                      "llm_compressor checkpoints")
        if not self.quantized:
            raise RuntimeError("call quantize() first")
        from .export import pack_block

        sym, bits = bool(self.scheme["sym"]), int(self.scheme["bits"])
        int_scheme = str(self.scheme["data_type"]).startswith("int")
        if format in ("auto_gptq", "auto_awq"):
            # the same scheme checks as the reference's format classes (auto_gptq.py:44-57, auto_awq.py:33-46)
            if not int_scheme or (self.scheme.get("act_bits") or 16) <= 8:
                raise ValueError(f"{format} format supports weight-only INT schemes, got data_type={self.scheme['data_type']} "
                                 f"act_bits={self.scheme.get('act_bits')}")
            if format == "auto_awq" and any(int(c.get("bits", 16)) not in (4, 16) for c in self.layer_config.values()):
                raise ValueError("auto_awq format supports W4A16 only (every tuned layer must be 4 bit)")
            backend = format
        elif format == "llm_compressor":      # same tensors as below, compressed-tensors config (export_to_llmcompressor/export_to_fp.py)
            if int_scheme:
                raise NotImplementedError("llm_compressor format for INT schemes packs through the compressed-tensors package "
                                          "(export_to_llmcompressor/export.py:209-246), which is not part of this path")
            llmc_cfg = self._llmc_quantization_config()
            backend = "llm_compressor"
        elif not int_scheme:     # MXFP4 / NVFP4 checkpoints carry the llm_compressor tensor layout (export_to_nvfp_mx.py:178-179)
            backend = "auto_round:llm_compressor"
        else:                    # AutoRoundFormat's defaults (export/formats/backends/autoround.py:59-70)
            backend = format if ":" in format else ("auto_round:auto_gptq" if sym else
                                                    ("auto_round:auto_awq" if bits == 4 else "auto_round"))
        # block-sharded runs: every rank packs (HIP packers) and writes the blocks IT tuned into its own shard files, rank 0 adds the
        # rest of the model, the merged index and the configs; data-parallel runs hold identical models everywhere: rank 0 writes
        multi = self.world > 1
        if multi and not self.sharded and self.rank != 0:
            _barrier()
            return None
        writer = ShardWriter(output_dir, max_shard_bytes=max_shard_bytes, tag=f"rank{self.rank}" if (multi and self.sharded) else None)
        packed_prefixes = []
        mine = set(getattr(self, "owned_blocks", range(len(self.block_names))))
        for bi, name in enumerate(self.block_names):
            block = self.model.get_submodule(name)
            if multi and self.sharded and bi not in mine:       # another rank's block: only its tensor names are needed here
                for ln, m in block.named_modules():
                    if hasattr(m, "scale") and int(getattr(m, "bits", 16)) < 16:
                        ln = ln[:-len(".orig_layer")] if ln.endswith(".orig_layer") else ln
                        packed_prefixes.append(f"{name}.{ln}.")
                continue
            packed = pack_block(block, backend if int_scheme else None)
            writer.write_block(name, packed)
            for ln in packed:
                ln = ln[:-len(".orig_layer")] if ln.endswith(".orig_layer") else ln
                packed_prefixes.append(f"{name}.{ln}.")
        if multi and self.sharded:
            import torch.distributed as dist

            if self.rank != 0:
                part = writer.finish()
                dist.gather_object(part, None, dst=0)
                _barrier()
                return None
        rest = {}
        tied = bool(getattr(getattr(self.model, "config", None), "tie_word_embeddings", False))
        emb = self.model.get_input_embeddings() if hasattr(self.model, "get_input_embeddings") else None
        for k, v in self.model.state_dict().items():
            k2 = k.replace(".orig_layer.", ".")
            if any(k2.startswith(p) for p in packed_prefixes):
                continue
            if tied and emb is not None and k2 == "lm_head.weight" and v.data_ptr() == emb.weight.data_ptr():
                continue        # tied output embedding: stored once, like save_pretrained does
            rest[k2] = v.detach().to("cpu").contiguous()
        writer.write(rest)
        if multi and self.sharded:
            import torch.distributed as dist

            parts = [None] * self.world
            dist.gather_object(writer.finish(), parts, dst=0)
            index = ShardWriter.write_index(output_dir, parts)
        else:
            index = writer.close()
        cfg = self.model.config.to_dict() if hasattr(self.model, "config") else {}
        if format == "llm_compressor":
            cfg["quantization_config"] = llmc_cfg
        elif format == "auto_gptq":
            cfg["quantization_config"] = self._gptq_quantization_config()
        elif format == "auto_awq":
            cfg["quantization_config"] = self._awq_quantization_config()
        else:
            cfg["quantization_config"] = self._quantization_config(backend)
        if format in ("auto_gptq", "auto_awq"):     # both exporters pin the advertised dtype to fp16 for the consumers' kernels
            cfg["torch_dtype"] = "float16"          # (export_to_autogptq/export.py:314, export_to_awq/export.py:217-219)
        with open(os.path.join(output_dir, "config.json"), "w") as f:
            json.dump(cfg, f, indent=2, default=str)
        with open(os.path.join(output_dir, "quantization_config.json"), "w") as f:      # the reference writes both
            json.dump(cfg["quantization_config"], f, indent=2, default=str)
        if self.tokenizer is not None and hasattr(self.tokenizer, "save_pretrained"):
            self.tokenizer.save_pretrained(output_dir)
        if multi:
            _barrier()
        return index

    def quantize_and_save(self, output_dir: str = "tmp_autoround", format: str = "auto_round", inplace: bool = True, **kw):
        model, _ = self.quantize()
        self.save_quantized(output_dir, format=format, inplace=inplace, **kw)
        return model, output_dir


# ---- one process per GPU ---------------------------------------------------------------------------------------------------------------
def dist_env():
    """(rank, world, local_rank) of this process: torch.distributed's if a group exists, else the launcher's environment
    (RANK / WORLD_SIZE / LOCAL_RANK as torchrun sets them), else (0, 1, 0)."""
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size(), int(os.environ.get("LOCAL_RANK", dist.get_rank()))
    world = int(os.environ.get("WORLD_SIZE", "1") or 1)
    rank = int(os.environ.get("RANK", "0") or 0)
    return rank, world, int(os.environ.get("LOCAL_RANK", rank) or 0)


def parse_device_map(device_map) -> List[torch.device]:
    """The reference's `device_map` values this path understands (autoround.py:750-757, utils/device.py:923-1040): an int, "cuda:1",
    "0,1,2,3" / [0, 1, 2, 3] (one device per rank), None / "auto" / "cuda" (device 0, or LOCAL_RANK's under a launcher)."""
    if device_map in (None, "auto", "cuda"):
        return [torch.device("cuda", 0)]
    if isinstance(device_map, int):
        return [torch.device("cuda", device_map)]
    if isinstance(device_map, torch.device):
        return [device_map]
    if isinstance(device_map, (list, tuple)):
        return [d for x in device_map for d in parse_device_map(x)]
    if isinstance(device_map, str):
        parts = [p.strip() for p in device_map.split(",") if p.strip()]
        if len(parts) > 1:
            return [d for x in parts for d in parse_device_map(x)]
        p = parts[0] if parts else "0"
        return [torch.device("cuda", int(p))] if p.isdigit() else [torch.device(p)]
    raise TypeError(f"device_map={device_map!r}: an int, a device string, or a comma separated list / a list of them")


def pick_rank_device(devices: List[torch.device], local_rank: int, world: int, device_count: int) -> torch.device:
    """This rank's device: the `local_rank`-th entry of a multi-device map; with ONE entry and several ranks the launcher's
    convention (cuda:LOCAL_RANK) where the node has that many devices, else the entry itself (several ranks sharing one GPU)."""
    if len(devices) > 1:
        return devices[local_rank % len(devices)]
    d = devices[0]
    if world > 1 and d.type == "cuda" and d.index in (None, 0) and local_rank < device_count:
        return torch.device("cuda", local_rank)
    return d


def init_process_group(device: torch.device):
    """torch.distributed's default group for a front-door run (backend "nccl" = RCCL over xGMI on ROCm; `AR_DIST_BACKEND=gloo` for
    ranks that share one GPU, which RCCL refuses); rendezvous from the launcher's MASTER_ADDR / MASTER_PORT (127.0.0.1 by default)."""
    import torch.distributed as dist

    if dist.is_initialized():
        return
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    backend = os.environ.get("AR_DIST_BACKEND", "nccl")
    torch.cuda.set_device(device)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=device)
    else:
        dist.init_process_group(backend)


def _barrier():
    import torch.distributed as dist

    if dist.is_available() and dist.is_initialized():
        torch.cuda.synchronize()
        dist.barrier()


TUNED_ATTRS = ("scale", "zp", "weight_global_scale", "input_global_scale", "act_max", "act_scale")


@torch.no_grad()
def sync_tuned_blocks(blocks, world: int, device, policy: str = "round_robin") -> int:
    """After a block-sharded run every rank holds the tuned form of ITS blocks only.  Each owner broadcasts what tuning left in its
    blocks -- every tuned linear's fake-quant weight (device to device over RCCL) and the small per-layer results (`scale`, `zp`,
    global scales, activation maxima; pickled) plus which layers now sit in an activation-quant shell -- so that `quantize()` returns
    the whole tuned model on every rank, as the sequential run does.  The one exchange step of the sharded path besides the
    calibration broadcast and the fp relay: 2 bytes per weight, once.  -> number of layers received."""
    import torch.distributed as dist

    from .sharding import owner_of
    from .wrapper import WrapperWALayer, _set_module

    rank = dist.get_rank()
    n, got = len(blocks), 0
    for k, block in enumerate(blocks):
        own = owner_of(k, n, world, policy)
        if own == rank:
            tuned = [(name, m) for name, m in block.named_modules() if isinstance(m, torch.nn.Module) and hasattr(m, "scale")
                     and int(getattr(m, "bits", 16)) < 16 and not isinstance(m, WrapperWALayer)]
            meta = [{"name": name[:-len(".orig_layer")] if name.endswith(".orig_layer") else name, "shell": name.endswith(".orig_layer"),
                     "attrs": {a: (getattr(m, a).detach().cpu() if isinstance(getattr(m, a), torch.Tensor) else getattr(m, a))
                               for a in TUNED_ATTRS if hasattr(m, a)}} for name, m in tuned]
            box = [meta]
        else:
            box = [None]
        dist.broadcast_object_list(box, src=own)
        meta = box[0]
        if own != rank:
            block.to(device)
        for item in meta:
            m = block.get_submodule(item["name"])
            if own != rank:
                if isinstance(m, WrapperWALayer):
                    m = m.orig_layer
                for a, v in item["attrs"].items():
                    setattr(m, a, v)
                if item["shell"] and not isinstance(block.get_submodule(item["name"]), WrapperWALayer):
                    _set_module(block, item["name"], WrapperWALayer(m))
                got += 1
            elif isinstance(m, WrapperWALayer):
                m = m.orig_layer
            dist.broadcast(m.weight.data, src=own)
    return got


class pre_block_modules_on_cpu:
    """Context: every parameter and buffer of `model` that is NOT inside one of `blocks` sits on the CPU; on exit they are back where
    they were.

    Why: the reference captures the first block's inputs ON THE CPU -- `calibration()` sets `calibrate_on_cpu` whenever only one block
    list's inputs are wanted (calibration/llm.py:74-90: "calibrate only the embedding layer (also fast on CPU)"), i.e. always for a
    decoder-only LLM: embedding, positional / rotary tables and the mask are made by the host's kernels, and the forward stops at the first
    block.  The embedding lookup is the same on either device; the ROTARY TABLES are not: fp32 cos / sin of the host's libm, rounded to
    bf16, differ from the GPU's in 6 of 262 144 values at Mixtral-8x7B's shape -- enough that every target of the block differs in its
    last bits (`tools/gpu/r06_mixtral_targets_probe.py`: with CPU-made tables the reference-free flow's targets equal the reference's
    digest).  A pre-hook on the first block stops the forward before anything of a block runs, so the blocks stay on the GPU."""

    def __init__(self, model, blocks):
        self.model = model
        self.skip = {id(t) for b in blocks for t in list(b.parameters()) + list(b.buffers())}
        self.moved = []

    def __enter__(self):
        for mod in self.model.modules():
            for store in (mod._parameters, mod._buffers):
                for k, t in store.items():
                    if t is None or id(t) in self.skip or t.device.type == "cpu":
                        continue
                    self.moved.append((t, t.device) if store is mod._parameters else (mod, k, t.device))
                    if store is mod._parameters:
                        t.data = t.data.to("cpu")
                    else:
                        store[k] = t.to("cpu")
        return self

    def __exit__(self, *exc):
        for rec in self.moved:
            if len(rec) == 2:
                rec[0].data = rec[0].data.to(rec[1])
            else:
                rec[0]._buffers[rec[1]] = rec[0]._buffers[rec[1]].to(rec[2])
        self.moved = []
        return False


def calibration_attention_mask(input_ids: torch.Tensor) -> torch.Tensor:
    """The [batch, S] attention mask the reference's calibrator passes for a dataset that is not one of its named ones
    (calibration/llm.py:374-402): ones; where a sample ends in repeats of its last token those repeats and the last position are
    cleared; the last position of EVERY sample is cleared (so the mask is never all ones)."""
    am = torch.ones_like(input_ids, dtype=torch.long)
    bsz, seq = input_ids.shape
    for i in range(bsz):
        last, j, repeated = input_ids[i, -1], seq - 2, False
        while j >= 0 and input_ids[i, j] == last:
            repeated = True
            am[i, j] = 0
            j -= 1
        if repeated:
            am[i, -1] = 0
    am[:, -1] = 0
    return am


def loss_mask_ids(tokens: torch.Tensor, pad_token_id=None) -> torch.Tensor:
    """Token ids with -100 where a position must not enter the loss (reference: calibration/llm.py:341-360): pad tokens --
    matched by `pad_token_id`, or, without one, the trailing repeats of a sample's last token -- and every sample's last
    position (no next-token target)."""
    ids = tokens.clone()
    if pad_token_id is not None:
        ids[ids == pad_token_id] = -100
    else:
        for b in range(ids.shape[0]):
            last, j = tokens[b, -1], tokens.shape[1] - 2
            while j >= 0 and tokens[b, j] == last:
                ids[b, j] = -100
                j -= 1
    ids[:, -1] = -100
    return ids


def _block_layer_config(layer_config, block_name, block):
    """The user's `layer_config` keys are full layer names or patterns over them; apply_scheme works on the names inside one
    block, so resolve the keys on the full names (schemes.expand_layer_config) and strip the block prefix."""
    if not layer_config:
        return None
    full = [f"{block_name}.{n}" for n, m in block.named_modules() if is_quantizable(m)]      # nn.Linear and Conv1D (GPT-2)
    return {n[len(block_name) + 1:]: over for n, over in expand_layer_config(full, layer_config).items()}


def _to_device(v, device):
    if isinstance(v, torch.Tensor):
        return v.to(device)
    if isinstance(v, (tuple, list)):
        return type(v)(_to_device(x, device) for x in v)
    return v


def _first_sample(v, bs):
    """Shared block kwargs are captured once; tensors that carry the calibration batch dimension keep one row so that they
    broadcast over any tuning batch size (attention masks, position ids; rotary (cos, sin) tuples are handled element-wise)."""
    if isinstance(v, torch.Tensor):
        return v[:1].detach() if (v.dim() > 0 and v.shape[0] == bs and bs > 1) else v.detach()
    if isinstance(v, (tuple, list)):
        return type(v)(_first_sample(x, bs) for x in v)
    return v


# The literal in `_llmc_quantization_config` restates what compressed-tensors 0.10.x emitted for its NVFP4 / MXFP4 presets (the
# layout auto_round/export/export_to_llmcompressor/config.py:103-139 also hard-codes).  When the package is importable its own
# answer wins, so fields later versions add (scale_dtype / zp_dtype, another observer for NVFP4 inputs) reach the checkpoint
# exactly as they would from the reference (config.py:56-101 `initialize_quantization` + export_to_fp.py:362-378).
LLMC_LITERAL_PINNED_TO = "compressed-tensors 0.10"


def _compressed_tensors_config(preset: str, ignore, fmt: str):
    """-> the dict the reference would dump for a preset scheme on every Linear, or None when compressed-tensors is absent."""
    try:
        from compressed_tensors.quantization import QuantizationConfig, QuantizationStatus, preset_name_to_scheme
    except Exception:
        return None
    try:        # an installed version without the preset (KeyError), another constructor signature, ...: the pinned literal is used
        group = preset_name_to_scheme(preset, ["Linear"])
        cfg = QuantizationConfig(config_groups={"group_0": group}, kv_cache_scheme=None, quantization_status=QuantizationStatus.COMPRESSED,
                                 ignore=list(ignore))
        setattr(cfg, "format", fmt)
        out = cfg.to_dict()
    except Exception as e:  # noqa: BLE001
        import warnings

        warnings.warn(f"compressed-tensors is importable but did not produce the {preset} config ({e!r}); writing the literal pinned to "
                      f"{LLMC_LITERAL_PINNED_TO}")
        return None
    out["provider"] = "auto-round"
    try:        # which version wrote the dict goes to the log, not into quantization_config: the reference's export writes the
        import logging      # compressed-tensors dict + provider only, and a strict loader may refuse an unknown field

        import compressed_tensors

        logging.getLogger("auto_round_amd").info("llm_compressor quantization_config written by compressed-tensors %s",
                                                 getattr(compressed_tensors, "__version__", "unknown"))
    except Exception:  # noqa: BLE001
        pass
    return out
