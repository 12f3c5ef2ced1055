"""Host side of the per-block sign-gradient tuning loop, mirroring the reference's operator interface:

    SignRoundQuantizer.quantize_block(block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids=None)
        reference: auto_round/algorithms/quantization/sign_round/quantizer.py:311-552
    block_forward / IndexSampler / collect_best_params
        reference: auto_round/compressors/utils.py:109-172, :388-438, :205-217
    compress_block (reference fp forward -> quantize_block -> quantized-output forward)
        reference: auto_round/algorithms/composer.py:360-483 (steps 3, 4, 6)

What is different from the reference, by design (MI355X-first, device resident):
  * calibration inputs/targets are ONE contiguous [nsamples, seq, hidden] HBM tensor each; minibatches are gathered
    by an index kernel from a pre-uploaded index schedule (the Python `random` stream is consumed exactly as the
    reference's IndexSampler consumes it);
  * per iteration the quant work is two grouped launches for the whole block (qdq forward; fused backward +
    sign-SGD + best-snapshot + next forward) instead of ~30 eager kernels per layer;
  * the loss, the `total_loss < best_loss` decision and the best-parameter copy stay on the device: there is no
    `.item()` per batch -- one host sync per block (unless dynamic_max_gap early stopping is requested).
The block's GEMMs/attention run through PyTorch-ROCm (hipBLASLt / SDPA on MFMA).
"""
from __future__ import annotations

import contextlib
import os
import functools
import copy
import inspect
import random
from dataclasses import dataclass
from typing import Any, Dict, List, Optional, Sequence, Union

import torch

from . import ops
from .sign_sgd import SignSGD
from .wrapper import (SignRoundOptimizedWrapperLinear, _quantizable, check_to_quantized, unwrapper_block,
                      update_block_global_scale_if_needed, wrapper_block)

FLT_MAX = float(torch.finfo(torch.float32).max)


# ----------------------------------------------------------------------------------------------------------------------
# small mirrors of reference helpers
# ----------------------------------------------------------------------------------------------------------------------
class IndexSampler:
    """Cyclic shuffled minibatch sampler drawing from Python's GLOBAL `random` stream, exactly like the reference
    (auto_round/compressors/utils.py:388-438): shuffle at construction, reshuffle when fewer than `batch_size`
    indices remain.  `transformers.set_seed(seed)` / `random.seed(seed)` before the run fixes the schedule."""

    def __init__(self, nsamples: int, batch_size: int) -> None:
        if batch_size <= 0 or batch_size > nsamples:
            raise ValueError("batch_size must be > 0 and <= nsamples")
        self.nsamples = nsamples
        self.batch_size = batch_size
        self.index = 0
        self.indices = list(range(nsamples))
        random.shuffle(self.indices)

    def next_batch(self) -> List[int]:
        if self.index + self.batch_size > self.nsamples:
            random.shuffle(self.indices)
            self.index = 0
        batch = self.indices[self.index:self.index + self.batch_size]
        self.index += self.batch_size
        return batch


def block_forward(block, input_ids, input_others, amp=False, amp_dtype=torch.bfloat16, device=None, output_return_id=0):
    """reference: auto_round/compressors/utils.py:109-172 -- first positional parameter of block.forward receives the
    hidden states, everything else goes by keyword; tuple outputs are reduced to element `output_return_id`."""
    others = dict(input_others) if input_others else {}
    positional = others.pop("positional_inputs", None) or ()
    names = [p for p in inspect.signature(block.forward).parameters.keys() if p != "self"]
    first = names[0] if names else "hidden_states"
    if first not in others:
        others[first] = input_ids
    for i, val in enumerate(positional):
        if i + 1 < len(names) and names[i + 1] not in others:
            others[names[i + 1]] = val
    if amp:
        with torch.autocast(device_type="cuda", dtype=amp_dtype):
            out = block(**others)
    else:
        out = block(**others)
    if isinstance(out, (tuple, list)):
        out = out[output_return_id]
    return out


def collect_best_params(block, cache_device=None) -> Dict[str, Dict[str, torch.Tensor]]:
    """reference: auto_round/compressors/utils.py:205-217 -- deep copy of every wrapper's tunable parameters."""
    params = {}
    for n, m in block.named_modules():
        if hasattr(m, "orig_layer") and hasattr(m, "params"):
            params[n] = {k: (p.data.to(cache_device, copy=True) if cache_device is not None else p.data.clone())
                         for k, p in m.params.items()}
    return params


def stack_samples(samples: Union[torch.Tensor, Sequence[torch.Tensor]], device) -> torch.Tensor:
    """list of [1, S, H] (the reference's per-sample cache) or an [N, S, H] tensor -> contiguous [N, S, H] in HBM."""
    if isinstance(samples, torch.Tensor):
        return samples.to(device).contiguous()
    return torch.cat([s.to(device) for s in samples], dim=0).contiguous()


SHARED_CACHE_KEYS = ("position_ids", "cache_position", "position_embeddings", "cu_seqlens")      # utils/common.py:676


def _to_device(v, device):
    if isinstance(v, torch.Tensor):
        return v.to(device)
    if isinstance(v, tuple):
        return tuple(_to_device(t, device) for t in v)
    return v


def normalize_input_others(input_others, nsamples: int, device, shared_keys=SHARED_CACHE_KEYS):
    """-> (shared, per_sample).  The reference's front door hands `input_others` over the way its input cache stores them
    (algorithms/block_runner.py:368-422 `_select_batch`): the shared keys as a list of which the first entry serves every
    batch, every other tensor-valued key as a list with one [1, ...] entry per calibration sample that it concatenates per
    minibatch.  Here the per-sample entries become ONE resident [N, ...] tensor per key (rows are gathered per minibatch by
    index); a key whose entries are all equal -- the usual fixed-seqlen case: one mask for every sample -- keeps a single
    broadcastable row and costs nothing per iteration.  A dict without list values (the standalone front door) is returned
    as it is."""
    if not input_others or not any(isinstance(v, list) for k, v in input_others.items() if k != "positional_inputs"):
        return input_others, {}
    shared, per_sample = {}, {}
    for k, v in input_others.items():
        if k == "positional_inputs":
            shared[k] = v
        elif k in shared_keys:
            shared[k] = _to_device(v[0] if isinstance(v, list) and len(v) else (None if isinstance(v, list) else v), device)
        elif isinstance(v, list) and len(v) == nsamples and all(isinstance(t, torch.Tensor) for t in v):
            rows = torch.cat([t.to(device) for t in v], dim=0).contiguous()
            if rows.shape[0] != nsamples:
                raise ValueError(f"input_others[{k!r}]: expected one row per calibration sample, got {tuple(rows.shape)}")
            if nsamples > 1 and bool((rows == rows[:1]).all()):
                shared[k] = rows[:1]
            else:
                per_sample[k] = rows
        elif isinstance(v, list):
            shared[k] = _to_device(v[0], device) if len(v) == 1 else v
        else:
            shared[k] = _to_device(v, device)
    return shared, per_sample


def check_need_act_calibration(act_dynamic, act_data_type=None, act_bits=16) -> bool:
    """reference: compressors/utils.py:186-202 -- static activation quantisation needs an `act_max` per layer."""
    if act_bits is None or act_bits > 8:
        return False
    if act_dynamic is not None and not act_dynamic:
        return True
    return act_data_type is not None and "static" in act_data_type


def register_act_max_hooks(block) -> list:
    """Forward hooks that track every statically activation-quantised layer's input maximum in `module.act_max`
    (reference: AlgorithmComposer._register_act_max_hooks, composer.py:223-281): one running max over all calibration
    tokens for NVFP4 ([1] tensor), a running per-row-group max otherwise."""
    def collect(module, inp, out):
        x = inp[0] if isinstance(inp, (tuple, list)) else inp
        if x.numel() == 0:
            return
        adt = str(getattr(module, "act_data_type", None) or getattr(module, "data_type", ""))
        if adt.startswith("nv_fp"):
            m = x.detach().abs().max().to(torch.float32).reshape(1)
            module.act_max = m if not hasattr(module, "act_max") or module.act_max.numel() == 0 \
                else torch.max(m, module.act_max.to(m.device).max().reshape(1))
            return
        gs = int(module.act_group_size)
        gs = x.shape[-1] if gs in (-1, 0) or x.shape[-1] < gs else gs
        if x.shape[-1] % gs:
            raise NotImplementedError("act_max calibration with a padded activation group")
        m = x.detach().reshape(-1, gs).abs().max(dim=-1).values
        module.act_max = m if not hasattr(module, "act_max") or module.act_max.numel() == 0 \
            else torch.max(m, module.act_max.to(m.device))

    return [m.register_forward_hook(collect) for n, m in block.named_modules()
            if n and _quantizable(m) and check_to_quantized(m) and hasattr(m, "act_dynamic")
            and check_need_act_calibration(m.act_dynamic, getattr(m, "act_data_type", None), getattr(m, "act_bits", 16))]


def set_amax_for_uncalibrated_experts(block, attr_name="act_max") -> int:
    """MoE experts that received no calibration token have no `act_max`; give them the maximum over their sibling experts'
    same-named linear (reference: set_amax_for_all_moe_layers / set_amax_for_uncalibrated_experts, utils/model.py:2003-2143).
    Returns the number of layers filled in."""
    filled = 0
    for mod in block.modules():
        experts = getattr(mod, "experts", None)
        if experts is None:
            continue
        from .moe_unfuse import expert_children

        experts = expert_children(experts)      # ModuleList of experts or an unfused experts module (numbered children)
        if not experts:
            continue
        names = sorted({n for e in experts for n, c in e.named_children() if _quantizable(c)})
        for name in names:
            members = [getattr(e, name) for e in experts if hasattr(e, name)]
            vals = [getattr(m, attr_name).reshape(-1) for m in members if getattr(m, attr_name, None) is not None]
            if not vals:
                continue
            top = torch.max(torch.cat([v.to(vals[0].device) for v in vals])).reshape(1)
            for m in members:
                if getattr(m, attr_name, None) is None:
                    setattr(m, attr_name, top.clone())
                    filled += 1
    return filled


# keyword arguments of a decoder block that the reference's input cache keeps PER SAMPLE and its runner concatenates per minibatch
PER_SAMPLE_MASK_KEYS = ("attention_mask", "attn_mask", "mask", "encoder_attention_mask", "causal_mask")


@dataclass
class BlockContext:
    """reference: algorithms BlockContext -- only the fields the quantizer reads."""
    block_index: int = 0
    block_cnt: int = 1
    block_name: str = ""


@dataclass
class SignRoundConfig:
    """The fields of the reference's SignRoundConfig / QuantizationConfig that steer the hot path
    (auto_round/algorithms/quantization/sign_round/config.py:20-170)."""
    iters: int = 200
    lr: Optional[float] = None
    minmax_lr: Optional[float] = None
    lr_scheduler: Any = None
    momentum: float = 0.0
    enable_minmax_tuning: bool = True
    enable_norm_bias_tuning: bool = False
    gradient_accumulate_steps: int = 1
    not_use_best_mse: bool = False
    dynamic_max_gap: int = -1
    enable_quanted_input: bool = True
    batch_size: int = 8
    bits: Optional[int] = 4
    amp: bool = True
    amp_dtype: torch.dtype = torch.bfloat16
    fuse_next_forward: bool = True       # MI355X: emit iteration i+1's Wq from the fused backward kernel
    # SDPA backend priority for the block's attention.  On MI355X / ROCm 7.2 / torch 2.10 the AOTriton "efficient"
    # kernels run the causal 8x32x2048x128 forward+backward in 2.66 ms vs 5.14 ms for the "flash" ones
    # (tools/sdpa_probe.py), so they are tried first; "auto" leaves torch's own choice untouched.
    sdpa_backend: str = "efficient"
    # Split every minibatch over the ranks of torch.distributed's default group and all-reduce the block's weight-gradient
    # buffer once per iteration (RCCL): multi-GPU speed-up for ONE block, usable with quantised-input chaining where blocks
    # cannot be sharded.  The reference's counterpart is its experimental DDP mode (utils/distributed.py).
    data_parallel: bool = False
    dp_overlap: bool = True              # dense blocks: per-layer gradient buckets all-reduced while the backward pass continues
    # Run supported decoder blocks (Llama / Mistral / Qwen2 / Qwen3 family: RMSNorm, rotary embedding, SwiGLU MLP, optional q/k norms; OPT family: LayerNorm, ReLU MLP) through the fused
    # HIP block path (auto_round_amd/fused_block.py) instead of transformers' module code -- the MI355X counterpart of the
    # reference's torch.compile(block_forward) (utils/device.py:112-122, compressors/base.py:1177-1179).  Same arithmetic per op,
    # different bf16 rounding points inside the block (trajectory-level parity, like the reference's compiled path); blocks it
    # does not cover silently keep the generic path.  Off by default for the same reason enable_torch_compile is.
    fused_block: bool = False
    # Weight-gradient GEMMs (dY^T X, 27 % of a Llama-3-8B iteration through hipBLASLt) on the hand-written MFMA kernel
    # (csrc/ar_gemm.hip) for the shapes where it wins (fused_block.mfma_dw_pays); another GEMM engine = another fp32 summation
    # order, so -- like fused_block -- opt-in.  The front door's enable_torch_compile=True switches both on.
    mfma_dw_gemm: bool = False
    # Inside the fused block: the causal attention forward on the hand-written flash-attention kernel (csrc/ar_attn.hip) instead of
    # torch's SDPA (AOTriton); at head size 128 the backward stays the library's, fed with this kernel's output and log-sum-exp rows.
    flash_attention: bool = True
    # Head size 64 (OPT): the attention backward on the hand-written deterministic kernel (csrc/ar_attn_bwd.hip) instead of the
    # library's; needs flash_attention (it consumes that forward's log-sum-exp rows).
    flash_attention_bwd: bool = True
    # Inside the fused block: the input-gradient GEMMs dX = dY W of o / gate-up / down read a transposed copy of the fake-quant
    # weights (one csrc/ar_block.hip transpose per weight per iteration) -- both operands contiguous along the reduction is the
    # layout hipBLASLt's tuned gfx950 kernel covers (fused_block.FusedLlamaBlock.set_tn_dx).
    tn_dx_gemm: bool = True
    # One tuning iteration of a fused block as ONE captured hipGraph, replayed `iters` times: the minibatch indices, the learning
    # rates and the iteration counter come from device tables (ar_iter_begin), the loss / best-loss bookkeeping already lives on
    # the device, so nothing in the iteration needs the host.  For launch-bound blocks (OPT-125M: ~60 launches per 2 ms
    # iteration).  None = automatic: blocks of up to `hip_graph_max_weights` quantised weights whose loop qualifies
    # (_graph_eligible); True = whenever the loop qualifies; False = never.  Same kernels in the same order as the host-driven
    # loop: results are bit-identical.
    hip_graph: Optional[bool] = None
    # Measured on the MI355X box (profiles/r03_bench_default.json `opt125m`): at OPT-125M's block size (7.1 M weights, ~60 launches
    # of ~25 us each per iteration) a deep host queue already keeps the GPU fed and the replayed graph is the SLOWER form (1.59 vs
    # 1.49 ms per iteration: ~2 us between graph nodes against ~1.5 from the queue); the graph pays where kernels are shorter than
    # the host's launch cost, i.e. for smaller blocks.  Hence the automatic mode stops at 4 M weights; `hip_graph=True` forces it.
    hip_graph_max_weights: int = 4 * 1024 * 1024
    # Module path (and exact_rounding, whose attention is the module path's call): hand the block its shared keyword tensors (one
    # attention mask for every sample, ...) materialised at the minibatch's own row count -- what the reference's per-sample input
    # cache produces by concatenation (block_runner.py:368-422: an [8, 1, S, S] mask, not a broadcastable [1, 1, S, S]).  Same values;
    # the library's attention picks another kernel for a batch-broadcast mask and returns other last bits at some shapes: Llama-3-8B's
    # are unaffected, but OPT-125M's block parted from the reference for exactly this reason (round 5: with the mask materialised -- and
    # torch's deterministic-algorithms mode on, as the reference's constructor leaves it -- the reference-free flow reproduces the
    # reference's targets and results bit for bit there; Mixtral-8x7B's targets still differ in 0.7 % of the attention output's last
    # bits).  ON by default since round 5 -- results identical to the reference's come first; the cost is one
    # [batch, 1, S, S] copy per block forward (67 MB at 8 x 2048).
    materialise_shared_rows: bool = True
    # The package's own NO-GRAD block forwards (the fp forward that makes a block's targets, the quantised-output forward that feeds the
    # next block) call the library attention in its training-mode form: same bits, but reproducible -- the inference-mode forward of
    # torch 2.10 / ROCm 7.2 returns 16-32 wrong values on 0.1-1 % of its calls at OPT-125M's shape (attention.reproducible_sdpa_forward,
    # profiles/r06_sdpa_flake.json).  Each call signature is compared once against the inference form before it is trusted.
    reproducible_attention_forward: bool = True
    # Opt-in determinism on a library that is not deterministic: every attention forward of the tuning loop is issued twice and the two
    # results compared on the device (attention.verified_sdpa_forward); a block whose run saw a differing pair is restored to its fp
    # weights and tuned again (same minibatch schedule), up to two times.  Cost: one more attention forward per iteration (+13 % at
    # OPT-125M on the exact path); without it about one OPT-125M run in fifteen takes a corrupted step -- as the reference itself does.
    verify_attention_forward: bool = False
    # small device-side operations around every attention call of the tuning loop (attention.guarded_sdpa: "before", "after", "touch")
    sdpa_guard: str = ""
    # Llama-family blocks through first-party kernels that keep the MODULE PATH'S BITS (auto_round_amd/exact_block.py,
    # csrc/ar_exact.hip): eager torch's rounding points and reduction order in the elementwise kernels, the module path's GEMM
    # shapes plus whichever faster GEMM forms prove bit-equal on this GPU / software stack.  Verified against the module code on
    # one real minibatch per kind of block before it is used; where the proof fails the module path runs.  Takes precedence over
    # `fused_block` (whose rounding points differ): with this switch on a block is never tuned on a path that is not bit-identical
    # to the module path.  The proof is a one-off cost per kind of block (classes, every linear's shape and dtype, minibatch shape,
    # mask): about 50 forward + backward passes with clones of the weight gradients -- ~1.5 s at Llama-3-8B's block dimensions, ~10 s
    # at Llama-3-70B's -- after which every later block of that kind is a dictionary lookup.
    exact_rounding: bool = False
    # MODULE-PATH blocks (no exact form: Mixtral, Qwen3, ...): run the block's attention on csrc/ar_attn_exact.hip -- the library
    # attention's bits at less than half its time -- through a transformers attention function installed for the block's tuning run,
    # AFTER a proof on two real minibatches: block output and every weight gradient equal to the stock module path's, and the
    # attention outputs / q, k, v gradients equal to torch's own directly.  Where the proof fails (other shapes, other masks, another
    # library build) the stock attention stays.  Only read when `exact_rounding` is on (the switch that asks for the module path's bits).
    exact_attention: bool = True

    def __post_init__(self):
        if self.iters < 0:
            self.iters = 200
        self.lr_is_auto = self.lr is None and self.iters > 0
        self.minmax_lr_is_auto = self.minmax_lr is None
        if self.lr_is_auto:
            self.lr = self._lr_for_bits(self.bits)
        self.minmax_lr = self.minmax_lr or self.lr

    def _lr_for_bits(self, bits):
        if self.iters <= 0:
            return None
        if self.iters >= 1000 and bits is not None and bits <= 3:
            return 2.0 / self.iters
        return 1.0 / self.iters

    def compute_lr(self, bits):
        return self._lr_for_bits(bits) if self.lr_is_auto else self.lr

    def compute_minmax_lr(self, bits):
        return self.compute_lr(bits) if self.minmax_lr_is_auto else self.minmax_lr


# ----------------------------------------------------------------------------------------------------------------------
# the quantizer
# ----------------------------------------------------------------------------------------------------------------------
@contextlib.contextmanager
def _no_uninitialised_fill():
    """Behind the reference's front door (and this package's, which mirrors it) torch runs in deterministic-algorithms mode, whose
    default also FILLS every `torch.empty` with NaN (torch.utils.deterministic.fill_uninitialized_memory): a memset launch per scratch
    buffer of every iteration.  Nothing here reads memory it did not write, so the fill is switched off for the tuning loop and put
    back afterwards; the mode itself (which ops / library kernels run) is left exactly as the caller set it."""
    det = getattr(torch.utils, "deterministic", None)
    if det is None or not torch.are_deterministic_algorithms_enabled() or not getattr(det, "fill_uninitialized_memory", False) \
            or os.environ.get("AR_KEEP_NAN_FILL") == "1":      # (probe: keep the NaN fill -- a read of unwritten memory then shows as NaN)
        yield
        return
    det.fill_uninitialized_memory = False
    try:
        yield
    finally:
        det.fill_uninitialized_memory = True


class SignRoundQuantizer:
    """MI355X implementation of the reference's block quantizer contract (quantization/base.py:148-179):
    `quantize_block(...) -> best_params`; post-conditions identical to the reference (block unwrapped in place,
    each quantised layer's weight holds the qdq values and carries `scale` / `zp`)."""

    wrapper_block = staticmethod(wrapper_block)
    optimizer = SignSGD

    def __init__(self, config: Optional[SignRoundConfig] = None, device: Union[str, torch.device] = "cuda", **kwargs):
        self.config = config or SignRoundConfig(**kwargs)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("SignRoundQuantizer needs a HIP device; there is no CPU fallback")
        c = self.config
        if c.enable_norm_bias_tuning:
            raise NotImplementedError("enable_norm_bias_tuning is outside the MI355X hot path")
        self.last_stats: Dict[str, Any] = {}
        self.last_fused_block = False
        self.last_hip_graph = False
        self._fused_verdict: Dict[Any, bool] = {}        # block signature -> did the fused kernels agree with the module code
        self._exact_warned: set = set()
        self._exact_plans: Dict[Any, Any] = {}           # (block signature, minibatch shape) -> proven exact_rounding plan | False
        self._modattn_verdict: Dict[Any, Any] = {}       # module-path attention on the first-party kernels: proven? (exact_attention)
        self._attn_restore: list = []                    # (config object, its _attn_implementation) to put back after the block
        self.last_module_exact_attention = False
        self.last_module_attention_report: Optional[dict] = None
        self.last_exact = False
        self.last_exact_report: Optional[dict] = None
        self._graph_stream = None
        self._graph_pool = None
        self._last_graph = None

    # convenience accessors with the reference's attribute names
    @property
    def iters(self): return self.config.iters
    @property
    def lr(self): return self.config.lr
    @property
    def minmax_lr(self): return self.config.minmax_lr
    @property
    def enable_minmax_tuning(self): return self.config.enable_minmax_tuning
    @property
    def enable_quanted_input(self): return self.config.enable_quanted_input
    @property
    def not_use_best_mse(self): return self.config.not_use_best_mse
    @property
    def dynamic_max_gap(self): return self.config.dynamic_max_gap
    @property
    def gradient_accumulate_steps(self): return self.config.gradient_accumulate_steps

    def _sdpa_ctx(self, seq=None):
        """SDPA backend priority for one attention call of `seq` tokens (attention.backend_order: the efficient kernels first,
        except for the lengths where their backward is wrong in this torch build)."""
        pref = getattr(self.config, "sdpa_backend", "auto")
        if pref == "auto":
            return contextlib.nullcontext()
        from torch.nn.attention import sdpa_kernel

        from .attention import backend_order

        order = backend_order(pref, seq)
        try:
            return sdpa_kernel(order, set_priority=True)
        except TypeError:  # pragma: no cover  (older torch without set_priority)
            return sdpa_kernel(order)

    def _others_for(self, rows: int, input_others, consumer=None):
        """`materialise_shared_rows`: the block's shared keyword tensors at the minibatch's own row count (see SignRoundConfig).
        `consumer`: an exact block whose proven plan runs the attention on the first-party kernels (plan["attn"]) reads the mask's
        STRUCTURE, not its rows (ops.mask_structure, cached per tensor object): the shared one-row mask is handed over as it is -- no
        64 MB copy and no host read per iteration -- and the block materialises the rows itself should it ever fall back to torch's
        attention (`materialise_mask_rows`)."""
        if not (self.config.materialise_shared_rows and input_others):
            return input_others
        if consumer is not None and getattr(consumer, "exact", False) and (getattr(consumer, "plan", None) or {}).get("attn"):
            consumer.materialise_mask_rows = True
            return input_others
        if consumer is None and self._attn_restore:      # module path with the proven first-party attention installed (exact_attention)
            return input_others

        def mat(v):
            # (whatever the dtype: transformers may hand the sdpa path a BOOLEAN mask, and the reference's runner concatenates that too)
            if isinstance(v, torch.Tensor) and v.dim() >= 3 and v.shape[0] == 1 and rows > 1:
                return v.expand(rows, *v.shape[1:]).contiguous()
            if isinstance(v, tuple):
                return tuple(mat(t) for t in v)
            return v

        # exactly what the reference's runner concatenates per minibatch: the per-sample keys (the attention mask); its SHARED keys --
        # position_ids, position_embeddings, cache_position (utils/common.py:676) -- stay one row there and must stay one row here: a
        # [8, S, 128] cos / sin instead of [1, S, 128] changes the layout `q * cos` returns, and with it the attention kernel's bits
        # (found on the Mixtral block, profiles/r05_t3_mixtral_forward_compare_*.json)
        # selected BY NAME: the attention mask and its aliases are the per-sample keys of a decoder block (block_runner.py:368-422);
        # any other [1, ...] keyword tensor stays as it was handed over
        return {k: (mat(v) if k in PER_SAMPLE_MASK_KEYS else v) for k, v in input_others.items()}

    def block_forward(self, block, x, input_others):
        input_others = self._others_for(x.shape[0], input_others)
        with self._sdpa_ctx(x.shape[1] if x.dim() == 3 else None):
            return block_forward(block, x, input_others, amp=self.config.amp, amp_dtype=self.config.amp_dtype)

    # ------------------------------------------------------------------------------------------------------------------
    def quantize_block(self, block, fp_inputs, input_others, fp_outputs, q_inputs=None, block_ctx=None, input_ids=None,
                       **kwargs) -> dict:
        # the whole block is tuned with the quantizer's device current: torch's ops take the device from their tensors, the
        # C-ABI launches take the stream of their tensors' device (ops._launch) -- both agree for any `device=`
        from .attention import guarded_sdpa

        with torch.cuda.device(self.device), _no_uninitialised_fill(), guarded_sdpa(getattr(self.config, "sdpa_guard", "") or ""):
            try:
                if not getattr(self.config, "verify_attention_forward", False):
                    return self._quantize_block(block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids, **kwargs)
                return self._quantize_block_verified(block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids, **kwargs)
            finally:
                self._restore_module_attention()

    def _quantize_block_verified(self, block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids, **kwargs):
        """`verify_attention_forward`: tune, and tune again from the block's fp weights and the same `random` state when the attention
        forward was caught returning two different results for one call during the run."""
        import random
        import warnings

        from .attention import verified_sdpa_forward
        from .wrapper import WrapperWALayer, _set_module

        snap = {n: m.weight.detach().clone() for n, m in block.named_modules()
                if _quantizable(m) and check_to_quantized(m)}
        attrs = {n: {a: getattr(block.get_submodule(n), a) for a in ("scale", "zp") if hasattr(block.get_submodule(n), a)} for n in snap}
        rstate = random.getstate()
        flag = torch.zeros(1, dtype=torch.bool, device=self.device)
        retries = 0
        while True:
            flag.zero_()
            with verified_sdpa_forward(flag):
                best = self._quantize_block(block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids, **kwargs)
            if not bool(flag.item()) or retries >= 2:
                break
            retries += 1
            warnings.warn(f"verify_attention_forward: the library attention returned two different results for one call while "
                          f"{type(block).__name__} was tuned; restoring the block and tuning it again (retry {retries})")
            for n, m in list(block.named_modules()):          # activation-quant shells of the finished run go, the fp weights come back
                if isinstance(m, WrapperWALayer):
                    _set_module(block, n, m.orig_layer)
            for n, w in snap.items():
                m = block.get_submodule(n)
                m.weight.data.copy_(w)
                for a in ("scale", "zp"):
                    if a in attrs[n]:
                        setattr(m, a, attrs[n][a])
                    elif hasattr(m, a):
                        delattr(m, a)
            random.setstate(rstate)
        self.last_stats["attention_forward_retries"] = retries
        self.last_stats["attention_forward_unverified"] = bool(flag.item())
        return best

    def _quantize_block(self, block, fp_inputs, input_others, fp_outputs, q_inputs=None, block_ctx=None, input_ids=None,
                        **kwargs) -> dict:
        cfg = self.config
        device = self.device
        active_inputs = q_inputs if (q_inputs is not None and cfg.enable_quanted_input) else fp_inputs
        X = stack_samples(active_inputs, device)
        Y = stack_samples(fp_outputs, device)
        nsamples = X.shape[0]
        # the reference's per-sample lists (plugin mode) -> resident tensors; the standalone front door's dict passes through
        input_others, per_sample_others = normalize_input_others(input_others, nsamples, device)

        quantized_names, unquantized_names = self.wrapper_block(
            block, cfg.enable_minmax_tuning, cfg.enable_norm_bias_tuning, enable_torch_compile=False, device=device,
            iters=cfg.iters)
        arenas = getattr(block, "_ar_arenas", [])
        if not arenas or cfg.iters <= 0:
            unwrapper_block(block, {})
            return {}
        for a in arenas:
            for lyr in a.layers:
                lyr._mfma_dw = bool(cfg.mfma_dw_gemm)
        fused = None
        self.last_exact = False
        if cfg.exact_rounding and cfg.amp:
            fused = self._build_exact(block, arenas, input_others, per_sample_others, X, Y,
                                      min(cfg.batch_size, min(nsamples, cfg.batch_size * cfg.gradient_accumulate_steps)))
            self.last_exact = fused is not None
            if fused is None:      # loud, once per kind of block: the module path is bit-identical too, but ~20 % slower
                kind = type(block).__name__
                if kind not in self._exact_warned:
                    self._exact_warned.add(kind)
                    import warnings

                    rep = self.last_exact_report if isinstance(self.last_exact_report, dict) and not self.last_exact_report.get("usable", True) else None
                    warnings.warn(f"exact_rounding: {kind} runs on the module path ("
                                  + ("the proof against the module code failed: " + str({k: rep[k] for k in ('base_mismatches', 'errors') if k in rep})
                                     if rep else "no exact form for this kind of block / these inputs") + ")")
        elif cfg.fused_block and cfg.amp:
            from .fused_block import build_fused_block

            fused = build_fused_block(block, arenas, input_others, cfg.amp_dtype, sdpa_ctx=self._sdpa_ctx, use_mfma_dw=cfg.mfma_dw_gemm,
                                      tn_dx_gemm=cfg.tn_dx_gemm)
            if fused is not None:
                fused.flash_fwd = bool(cfg.flash_attention)
                fused.flash_bwd = bool(cfg.flash_attention_bwd)
                if cfg.momentum and getattr(fused, "grouped", False):
                    # the grouped weight-gradient launch writes a ZERO gradient for an expert that got no rows and marks it as
                    # computed; with momentum the step then decays that expert's buffer and applies sign(buf), where the reference's
                    # SignSGD skips parameters whose grad is None (sign_sgd.py:356-389) -- the per-expert loop keeps that distinction
                    fused.grouped = False
                # the fused kernels must compute what the block's own code computes: one small minibatch through both
                # (once per kind of block: the verdict is remembered by class and by whether any submodule carries its own forward)
                key = ("tune", self._block_signature(block), self._others_signature(block, input_others))
                if key not in self._fused_verdict:
                    nchk = min(2, nsamples)
                    others_chk = input_others if not per_sample_others else {**input_others, **{k: t[:nchk] for k, t in per_sample_others.items()}}
                    self._fused_verdict[key] = fused.agrees_with_module(lambda x, o: self.block_forward(block, x, o), X[:nchk], others_chk)
                    self._report_fused_verdict(block, fused, self._fused_verdict[key])
                if not self._fused_verdict[key]:
                    fused = None
        self.last_fused_block = fused is not None
        self.last_module_exact_attention = False
        if fused is None and cfg.exact_rounding and cfg.exact_attention and cfg.amp and not cfg.data_parallel:
            self.last_module_exact_attention = self._install_module_attention(
                block, arenas, input_others, per_sample_others, X, Y,
                min(cfg.batch_size, min(nsamples, cfg.batch_size * cfg.gradient_accumulate_steps)))

        # one (round, minmax) pair of param groups per arena; lr by the arena's bit-width (quantizer.py:374-417)
        groups = []
        for ai, a in enumerate(arenas):
            layer_lr = cfg.compute_lr(a.bits) or cfg.lr
            layer_mm = cfg.compute_minmax_lr(a.bits) or cfg.minmax_lr
            groups.append({"params": [l.params["value"] for l in a.layers if "value" in l.params],
                           "lr": torch.tensor(layer_lr), "kind": "round", "arena": ai})
            if cfg.enable_minmax_tuning:
                mm = []
                for l in a.layers:
                    mm += [p for k, p in l.params.items() if "min" in k or "max" in k]
                groups.append({"params": mm, "lr": torch.tensor(layer_mm), "kind": "minmax", "arena": ai})
        optimizer = self.optimizer(groups, lr=torch.tensor(cfg.lr), weight_decay=0, arenas=arenas, momentum=cfg.momentum or 0.0)
        optimizer.fuse_next_fwd = cfg.fuse_next_forward
        if cfg.lr_scheduler is None:
            lr_schedule = torch.optim.lr_scheduler.LinearLR(optimizer, start_factor=1.0, end_factor=0.0,
                                                            total_iters=cfg.iters)
        else:
            lr_schedule = copy.deepcopy(cfg.lr_scheduler)

        batch_size = cfg.batch_size
        global_bs = min(nsamples, batch_size * cfg.gradient_accumulate_steps)
        accum = cfg.gradient_accumulate_steps != 1
        sampler = IndexSampler(nsamples, global_bs) if kwargs.get("index_schedule") is None else None
        early_stop = (not cfg.not_use_best_mse) and cfg.dynamic_max_gap > 0
        # the whole index schedule is drawn up front (same draws, same order as one next_batch() per iteration) and
        # uploaded once; with early stopping the draws must stay lazy so later blocks see the reference's stream.
        index_schedule = kwargs.get("index_schedule")
        if index_schedule is not None:      # sharded runs replay the sequential run's schedule for this block
            sched_dev = torch.tensor(index_schedule, dtype=torch.int64).to(device, non_blocking=True)
            early_stop = False
        elif early_stop:
            sched_dev = None
        else:
            sched = [sampler.next_batch() for _ in range(cfg.iters)]
            sched_dev = torch.tensor(sched, dtype=torch.int64).to(device, non_blocking=True)

        # valid-token loss mask (reference: quantization/base.py:257-280): positions whose token id the calibrator set
        # to -100 (pads, the last token of every sample) are excluded from the loss; all-valid -> unmasked fast path
        mask_dev, valid_counts = None, None
        if input_ids is not None:
            ids = input_ids if isinstance(input_ids, torch.Tensor) else torch.cat([t.reshape(1, -1) for t in input_ids], 0)
            valid = ids.reshape(nsamples, -1) != -100
            if not bool(valid.all()):
                mask_dev = valid.to(torch.uint8).to(device).contiguous()
                valid_counts = valid.sum(dim=1).tolist()
                mb = torch.empty((min(batch_size, global_bs), mask_dev.shape[1]), dtype=torch.uint8, device=device)
        sched_host = index_schedule if index_schedule is not None else (None if sched_dev is None else sched)

        dp_rank, dp_size = 0, 1
        accum_cfg = cfg.gradient_accumulate_steps != 1     # with micro-batches a layer's gradient is only complete after the last one
        self.last_dp_overlapped = False
        if cfg.data_parallel:
            from .sharding import dp_world, sync_block_gradients

            dp_rank, dp_size = dp_world()
            if dp_size > 1 and cfg.dp_overlap and not accum_cfg:
                from .sharding import enable_overlapped_sync

                self.last_dp_overlapped = enable_overlapped_sync(block, arenas)
            if dp_size > 1 and (min(batch_size, global_bs) % dp_size or global_bs % min(batch_size, global_bs)):
                raise ValueError(f"data_parallel: batch_size {batch_size} (global {global_bs}) must be divisible by the "
                                 f"{dp_size} ranks so that every rank weighs the same in the summed gradient")
        total_loss = torch.zeros(1, dtype=torch.float32, device=device)
        state = torch.tensor([FLT_MAX, 0.0, 0.0], dtype=torch.float32, device=device)
        istate = torch.zeros(4, dtype=torch.int32, device=device)
        loss_hist = torch.zeros(max(cfg.iters, 1), dtype=torch.float32, device=device)     # per-iteration loss, read once at the end
        track_best = not cfg.not_use_best_mse
        if track_best:
            optimizer.snapshot_flag = istate[0:1]
        xb = torch.empty((min(batch_size, global_bs),) + tuple(X.shape[1:]), dtype=X.dtype, device=device)
        yb = torch.empty((min(batch_size, global_bs),) + tuple(Y.shape[1:]), dtype=Y.dtype, device=device)
        scratch = {"dpred": None}
        last_iter = cfg.iters - 1

        def run_minibatches(gidx, num_elm, direct=False):
            """forward / loss / backward of one iteration's minibatches; `gidx`: device int64 [global_bs] sample indices"""
            for b0 in range(0, global_bs, batch_size):
                idx = gidx[b0:b0 + batch_size]
                if dp_size > 1:     # this rank's share of the minibatch; the summed gradient is the full-batch one
                    idx = idx[dp_rank::dp_size].contiguous()
                nb = idx.numel()
                x = ops.gather_rows(X, idx, out=xb[:nb])
                ref = ops.gather_rows(Y, idx, out=yb[:nb])
                others_b = input_others
                if per_sample_others:       # rows of this minibatch, like the reference's per-batch concatenation
                    others_b = {**input_others, **{k: t.index_select(0, idx) for k, t in per_sample_others.items()}}
                ctx = None
                if direct:          # captured iterations: the fused block's own forward / backward, no autograd graph in between
                    pred, ctx = fused.forward_direct(x, others_b, donate_input=True)      # (exact blocks are never captured: capturable = False)
                else:
                    if fused is not None:
                        pred = fused.forward(x, self._others_for(nb, others_b, consumer=fused) if self.last_exact else others_b, donate_input=True)
                    else:
                        pred = self.block_forward(block, x, others_b)      # x: scratch rows
                pred_c = pred if pred.is_contiguous() else pred.contiguous()
                dpred = scratch["dpred"]
                if dpred is None or dpred.shape != pred_c.shape or dpred.dtype != pred_c.dtype:
                    dpred = scratch["dpred"] = torch.empty_like(pred_c)
                n = pred_c.numel()
                tmask = None
                if mask_dev is not None:
                    if mask_dev.shape[1] % 16 == 0:
                        tmask = ops.gather_rows(mask_dev, idx, out=mb[:nb]).view(-1)
                    else:
                        tmask = mask_dev.index_select(0, idx).contiguous().view(-1)
                self._loss_fwd_bwd(pred_c, ref.to(pred_c.dtype), dpred, total_loss, n, num_elm, accum, tmask)
                if direct:
                    fused.backward_direct(ctx, dpred)
                else:
                    pred_c.backward(dpred)

        use_graph = self._graph_eligible(cfg, fused, arenas, early_stop, dp_size, accum, per_sample_others, valid_counts, sched_dev, track_best)
        self.last_hip_graph = False
        first_host_iter = 0
        if use_graph:
            first_host_iter = self._run_captured(cfg, optimizer, lr_schedule, arenas, sched_dev, global_bs, valid_counts, run_minibatches,
                                                 total_loss, state, istate, loss_hist, device)

        for i in range(first_host_iter, cfg.iters):
            if sched_dev is None:
                host_idx = sampler.next_batch()
                gidx = torch.tensor(host_idx, dtype=torch.int64).to(device)
            else:
                host_idx = sched_host[i]
                gidx = sched_dev[i]
            num_elm = 1
            if valid_counts is not None:   # number of valid tokens in the global batch (quantizer.py:477-478)
                num_elm = max(1, sum(valid_counts[j] for j in host_idx))
            elif accum:
                num_elm = global_bs * X[0].numel()
            run_minibatches(gidx, num_elm)
            if dp_size > 1:
                sync_block_gradients(arenas, total_loss, average_loss=not accum)
            ops.best_loss_update(total_loss, state, istate, i, loss_hist=loss_hist)
            if early_stop:
                last_best = int(istate[1].item())
                if 0 < cfg.dynamic_max_gap <= i - last_best:
                    last_iter = i
                    break
            if not track_best and i == cfg.iters - 1:
                # not_use_best_mse: the reference snapshots the parameters at the last iteration BEFORE its optimizer step
                # (sign_round/quantizer.py:513-514), so that step never reaches the baked weights -- do not take it
                optimizer.zero_grad()
                break
            optimizer.step()
            optimizer.zero_grad()
            lr_schedule.step()

        # one host sync per block
        st = state.tolist()
        ist = istate.tolist()
        best_loss, init_loss, last_loss = st
        n_improved = ist[2]
        best_params: Dict[str, Dict[str, torch.Tensor]] = {}
        if track_best:
            if n_improved > 0:
                for a in arenas:
                    for l in a.layers:
                        best_params[_name_of(block, l)] = a.best_params_of(l)
            best_iter, shown_loss = ist[1], best_loss
        else:
            best_params = collect_best_params(block)
            best_iter, shown_loss = cfg.iters, last_loss
        self.last_stats = dict(init_loss=init_loss, best_loss=best_loss, last_loss=last_loss, best_iter=best_iter,
                               iters_run=last_iter + 1, quantized=len(quantized_names),
                               unquantized=len(unquantized_names), n_improved=n_improved, hip_graph=bool(self.last_hip_graph),
                               loss_trace=loss_hist[:last_iter + 1].tolist())
        if cfg.exact_rounding:      # which path really ran, and with which proven forms (flat: the plugin's log line prints them)
            rep = (self.last_exact_report or {}) if self.last_exact else {}
            self.last_stats.update(exact_block=bool(self.last_exact), exact_plan=rep.get("plan"), exact_streamk=rep.get("streamk"),
                                   exact_dropped=rep.get("dropped") or None, exact_kept_on_second_try=rep.get("kept_on_second_try"))
        with torch.no_grad():
            unwrapper_block(block, best_params)
        # hand out independent copies: the arenas' best_* buffers are released with the block's wrappers
        return {n: {k: v.clone() for k, v in d.items()} for n, d in best_params.items()}

    # -- module path: the block's attention on the first-party kernels (exact_attention) ------------------------------------------
    def _restore_module_attention(self):
        from . import attention as A

        while self._attn_restore:
            cfg_obj, old = self._attn_restore.pop()
            cfg_obj._attn_implementation = old
        A.exact_state["materialise"] = False
        A.exact_state["key_block"] = 0

    def _install_module_attention(self, block, arenas, input_others, per_sample_others, X, Y, rows) -> bool:
        """Proof, then installation for this block's tuning run: the module code with transformers' attention function swapped for
        `attention.exact_sdpa_attention` must return the stock module path's block output and weight gradients bit for bit on two
        real minibatches, with every attention call's output and q / k / v gradients equal to torch's own (compared inside the call).
        The verdict is remembered per kind of block and minibatch shape."""
        import warnings

        from . import attention as A
        from .exact_block import _count_diff

        mask = input_others.get("attention_mask") if isinstance(input_others, dict) else None
        cfgs = {}
        for m in block.modules():
            c = getattr(m, "config", None)
            if c is not None and getattr(c, "_attn_implementation", None) == "sdpa":
                cfgs[id(c)] = c
        if mask is None or not cfgs or per_sample_others or X.shape[0] < rows:
            return False
        key = ("modattn", self._block_signature(block), self._shape_signature(block), rows, tuple(X.shape[1:]), str(X.dtype),
               (tuple(mask.shape), str(mask.dtype)), self.config.sdpa_backend, self.config.materialise_shared_rows)
        verdict = self._modattn_verdict.get(key)
        name = A.register_exact_sdpa()

        def install(key_block=0):
            for c in cfgs.values():
                self._attn_restore.append((c, c._attn_implementation))
                c._attn_implementation = name
            A.exact_state["materialise"] = bool(self.config.materialise_shared_rows)
            A.exact_state["key_block"] = int(key_block)

        if verdict is not None:
            if verdict:
                install(verdict if isinstance(verdict, int) and not isinstance(verdict, bool) else 0)
            return bool(verdict)
        for a in arenas:
            if not a.wq_fresh:
                a.qdq_forward()
        reset = lambda: [l._dw_accum.__setitem__(0, False) for a in arenas for l in a.layers]  # noqa: E731

        def run(lo):
            reset()
            xb = X[lo:lo + rows].clone()
            pred = self.block_forward(block, xb, input_others)
            pred_c = pred if pred.is_contiguous() else pred.contiguous()
            dpred = torch.empty_like(pred_c)
            scratch = torch.zeros(1, dtype=torch.float32, device=xb.device)
            ops.mse_loss_fwd_bwd(pred_c, Y[lo:lo + rows].to(pred_c.dtype), dpred=dpred, loss_accum=scratch, accum_scale=1.0, grad_scale=1000.0)
            pred_c.backward(dpred)
            return pred_c.detach(), [a.dWq.clone() for a in arenas]

        los = [0] + ([rows] if X.shape[0] >= 2 * rows else [])
        report = dict(minibatches=len(los))
        try:
            refs = [run(lo) for lo in los]
            verdict = False
            # the library picks its forward configuration by shape: the measured guess (0) first, then the other key blocks
            for kb in (0, 64, 32, 16):
                self._restore_module_attention()
                install(kb)
                A.exact_state.update(verify=True, diffs={}, calls=0, fallbacks=0)
                n_bad = 0
                for lo, (y_ref, dw_ref) in zip(los, refs):
                    y, dws = run(lo)
                    n_bad += _count_diff(y, y_ref) + sum(_count_diff(a, b) for a, b in zip(dws, dw_ref))
                direct = dict(A.exact_state["diffs"])
                report.update(block_mismatches=n_bad, attn_direct=direct, calls=A.exact_state["calls"], fallbacks=A.exact_state["fallbacks"],
                              key_block=kb)
                if (n_bad == 0 and A.exact_state["calls"] > 0 and A.exact_state["fallbacks"] == 0
                        and set(direct) == {"out", "dq", "dk", "dv"} and not any(direct.values())):
                    verdict = kb if kb else True
                    break
                if A.exact_state["calls"] == 0:      # no call the kernels take: other key blocks will not change that
                    break
        except Exception as e:  # noqa: BLE001 -- never an aborted run: the stock attention stays
            report["error"] = repr(e)[:200]
            verdict = False
        finally:
            A.exact_state["verify"] = False
            reset()
        self._modattn_verdict[key] = verdict
        self.last_module_attention_report = dict(report, usable=bool(verdict))
        if not verdict:
            self._restore_module_attention()
            warnings.warn(f"exact_attention: {type(block).__name__} keeps torch's own attention on the module path ({report})")
        return bool(verdict)

    def _build_exact(self, block, arenas, input_others, per_sample_others, X, Y, rows: int):
        """The exact_rounding form of a wrapped block, or None (module path): recognised by exact_block.ExactLlamaBlock and PROVEN
        bit-equal to the module code on one real minibatch (first `rows` samples) -- once per kind of block and minibatch shape;
        later blocks of the same kind reuse the plan."""
        from .exact_block import ExactLlamaBlock
        from .exact_opt_block import ExactOPTBlock

        cfg = self.config
        if cfg.data_parallel or not isinstance(input_others, dict):
            return None
        eb = None
        for cls in (ExactLlamaBlock, ExactOPTBlock):        # Llama family (round 4), OPT family (round 6: BASELINE configs[0])
            eb = cls.try_build(block, arenas, input_others, cfg.amp_dtype, sdpa_ctx=self._sdpa_ctx, amp=cfg.amp)
            if eb is not None:
                break
        if eb is None:
            return None
        mask = input_others.get("attention_mask")
        key = ("exact", self._block_signature(block), self._shape_signature(block), rows, tuple(X.shape[1:]), str(X.dtype),
               bool(per_sample_others), None if mask is None else (tuple(mask.shape), str(mask.dtype)), cfg.sdpa_backend,
               cfg.materialise_shared_rows)
        plan = self._exact_plans.get(key)
        if plan is None:
            def others_of(lo):
                o = input_others if not per_sample_others else {**input_others, **{k: t[lo:lo + rows] for k, t in per_sample_others.items()}}
                return self._others_for(rows, o)

            # a second minibatch of the same shape for the confirming run of every option (exact_block.plan_against_module)
            second = (X[rows:2 * rows].clone(), others_of(rows), Y[rows:2 * rows]) if X.shape[0] >= 2 * rows else None
            try:
                plan = eb.plan_against_module(lambda x, o: self.block_forward(block, x, o), X[:rows].clone(), others_of(0), Y[:rows],
                                              second=second)
            except Exception as e:  # noqa: BLE001 -- a block the class does not fit after all: the module path, never an aborted run
                import warnings

                warnings.warn(f"exact_rounding: the proof against the module code raised {e!r}; {type(block).__name__} keeps the module path")
                plan = None
                for a in arenas:
                    for l in a.layers:
                        l._dw_accum[0] = False
            self._exact_plans[key] = plan if plan is not None else False
            self.last_exact_report = eb.plan_report
        if not plan:
            return None
        if cfg.gradient_accumulate_steps != 1:
            # micro-batches accumulate dW with addmm_ in the module path (the proof covered the plain product): every weight gradient
            # through the library exactly as _QLinearFn.backward issues it, layer by layer
            plan = {k: (0 if k.startswith("dw_") else v) for k, v in plan.items()}
        eb.set_plan(plan)
        return eb

    def _build_exact_plain(self, block, inputs, input_others, rows: int):
        """The exact_rounding form of an UNWRAPPED block for the no-grad passes (targets, quantised-output forward), proven against the
        module code's output bits once per kind of block; None: the module path."""
        from .exact_block import ExactLlamaBlock
        from .exact_opt_block import ExactOPTBlock

        cfg = self.config
        eb = None
        for cls in (ExactLlamaBlock, ExactOPTBlock):
            try:
                eb = cls.try_build_plain(block, input_others, cfg.amp_dtype, sdpa_ctx=self._sdpa_ctx, amp=cfg.amp)
            except Exception:  # noqa: BLE001 -- anything unexpected about the block: the module path
                eb = None
            if eb is not None:
                break
        if eb is None:
            return None
        mask = input_others.get("attention_mask")
        key = ("exact_plain", self._block_signature(block), self._shape_signature(block), rows, tuple(inputs.shape[1:]), str(inputs.dtype),
               None if mask is None else (tuple(mask.shape), str(mask.dtype)), cfg.sdpa_backend, cfg.materialise_shared_rows)
        plan = self._exact_plans.get(key)
        if plan is None:
            try:
                plan = eb.plan_forward_against_module(lambda x, o: self.block_forward(block, x, o), inputs[:rows],
                                                      self._others_for(rows, input_others))
            except Exception as e:  # noqa: BLE001
                import warnings

                warnings.warn(f"exact_rounding (no-grad form): the proof against the module code raised {e!r}; module path")
                plan = None
            self._exact_plans[key] = plan if plan is not None else False
        if not plan:
            return None
        eb.set_plan(plan)
        return eb

    @staticmethod
    def _others_signature(block, input_others):
        """what else changes the arithmetic between two blocks of one class: whether a mask is among the inputs (the attention then
        takes another kernel) and per-layer attributes such as the layer type of hybrid (sliding / full attention) stacks"""
        mask = (input_others or {}).get("attention_mask") if isinstance(input_others, dict) else None
        return (mask is not None, str(getattr(block, "attention_type", getattr(getattr(block, "self_attn", None), "layer_type", ""))),
                getattr(getattr(block, "self_attn", None), "sliding_window", None) is not None)

    @staticmethod
    def _report_fused_verdict(block, fused, ok):
        import warnings

        from .fused_block import LLAMA_FAMILY, MOE_FAMILY, OPT_FAMILY

        d = getattr(fused, "last_disagreement", None)
        if not ok and type(block).__name__ in LLAMA_FAMILY + OPT_FAMILY + MOE_FAMILY:
            warnings.warn(f"fused block path: {type(block).__name__} is on the class whitelist but its fused form does not agree with the "
                          f"module code on the check minibatch (distance / block contribution = {d}); every block of this kind keeps "
                          f"the module path")

    @staticmethod
    def _shape_signature(block):
        """What a PROVEN exact_rounding plan is specific to besides the block's classes: every linear's (name, out, in, weight dtype,
        bias) and the attention's head geometry.  The plan holds shape-specific GEMM forms (`dw_*` = n K-slices or a stream-K cut table
        found for ONE (M, N, K); `tn_*`), and hipBLASLt picks its kernel by shape: a later block of the same class with another FFN or
        KV width (variable-width Llama derivatives) must get its own proof, not inherit one."""
        lins = []
        for n, m in block.named_modules():
            w = getattr(m, "weight", None)
            if isinstance(w, torch.Tensor) and w.dim() == 2 and (isinstance(m, torch.nn.Linear) or hasattr(m, "orig_layer")):
                lins.append((n.replace(".orig_layer", ""), tuple(w.shape), str(w.dtype), getattr(m, "bias", None) is not None))
        att = getattr(block, "self_attn", None)
        geo = tuple(getattr(att, k, None) if not isinstance(getattr(att, k, None), torch.Tensor) else None
                    for k in ("head_dim", "num_key_value_groups", "scaling", "num_heads", "num_key_value_heads")) if att is not None else ()
        acfg = getattr(att, "config", None)
        geo += tuple(getattr(acfg, k, None) for k in ("num_attention_heads", "num_key_value_heads", "head_dim")) if acfg is not None else ()
        return (tuple(sorted(set(lins))), geo)

    @staticmethod
    def _block_signature(block):
        """What decides whether a fused form computes the block's function: the classes involved, the scheme-relevant switches and
        whether any submodule carries an instance-level `forward` (a patched module)."""
        mods = list(block.modules())
        return (tuple(sorted({type(m).__name__ for m in mods})), any("forward" in m.__dict__ for m in mods),
                tuple(sorted({(int(getattr(m, "act_bits", 16) or 16), str(getattr(m, "act_data_type", ""))) for m in mods if hasattr(m, "act_bits")})))

    # -- one iteration as a captured hipGraph -------------------------------------------------------------------------------
    @staticmethod
    def _graph_eligible(cfg, fused, arenas, early_stop, dp_size, accum, per_sample_others, valid_counts, sched_dev, track_best) -> bool:
        """The loop can run from device tables when nothing inside an iteration depends on the host: a fused block (its forward /
        backward are plain kernel sequences), the whole index schedule drawn up front, best-parameter tracking on the device, a
        valid-token count that is the same for every minibatch (it is a kernel ARGUMENT of the loss), no early stopping, no
        micro-batches, no data-parallel exchange, no momentum buffers, no per-tensor (shared-parameter) arenas."""
        if cfg.hip_graph is False or fused is None or not track_best or early_stop or dp_size > 1 or accum or per_sample_others:
            return False
        if not getattr(fused, "capturable", True):      # sparse-MoE blocks read the per-expert token counts on the host
            return False
        if sched_dev is None or cfg.iters < 3 or (cfg.momentum or 0.0) or any(a.shared for a in arenas):
            return False
        if valid_counts is not None and len(set(valid_counts)) != 1:
            return False
        if cfg.hip_graph is None and sum(a.n for a in arenas) > cfg.hip_graph_max_weights:
            return False        # big blocks are GPU-bound (19 ms of kernels per Llama-3-8B iteration): nothing to gain
        return True

    def _run_captured(self, cfg, optimizer, lr_schedule, arenas, sched_dev, global_bs, valid_counts, run_minibatches, total_loss, state,
                      istate, loss_hist, device) -> int:
        """Iteration 0 eagerly through the device-table form (allocates every scratch buffer, sets the kernels' attributes, leaves
        the Python-side freshness flags in their steady state), capture of ONE iteration, then iters - 1 replays.  -> the first
        iteration the host-driven loop still has to run (== iters when everything was replayed)."""
        import warnings

        iters = cfg.iters
        # the scheduler's learning-rate sequence, computed by the scheduler itself (chained fp32 recurrence of LinearLR on the
        # 0-dim lr tensors): value used by iteration i = value after i scheduler steps
        rows = []
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for _ in range(iters):
                row = []
                for a in arenas:
                    lr_v, lr_mm = optimizer._arena_lrs(a)
                    row += [float(lr_v), float(lr_mm)]
                rows.append(row)
                lr_schedule.step()
        lr_table = torch.tensor(rows, dtype=torch.float32).t().contiguous().to(device)            # [2 * arenas, iters]
        lr_out = torch.zeros(2 * len(arenas), dtype=torch.float32, device=device)
        optimizer.lr_override = {id(a): (lr_out[2 * k:2 * k + 1], lr_out[2 * k + 1:2 * k + 2]) for k, a in enumerate(arenas)}
        it_dev = torch.zeros(1, dtype=torch.int32, device=device)
        cur_idx = torch.empty(global_bs, dtype=torch.int64, device=device)
        sched_flat = sched_dev.reshape(-1).contiguous()
        num_elm = 1 if valid_counts is None else max(1, valid_counts[0] * global_bs)

        def body():
            ops.iter_begin(it_dev, sched_flat, cur_idx, lr_table, lr_out, iters)
            run_minibatches(cur_idx, num_elm, direct=True)
            ops.best_loss_update(total_loss, state, istate, 0, iter_dev=it_dev, loss_hist=loss_hist)
            optimizer.step()

        try:
            body()                                                      # iteration 0
        except BaseException:
            optimizer.lr_override = None
            raise
        flags = [(a, a.wq_fresh, [l._dw_accum[0] for l in a.layers]) for a in arenas]
        prev_prof = ops.profile_enable(False)                           # per-dispatch event pairs cannot be captured
        graph = None
        try:
            # (capture_begin / capture_end on a side stream directly: the torch.cuda.graph() context also runs gc.collect() and
            #  empties the allocator's cache on entry -- tens of milliseconds per block on a 380 ms block)
            # One allocator pool for the graphs of all blocks: a fresh private pool per capture means fresh hipMallocs for every
            # activation of every block (tens of milliseconds on a 380 ms block).  The previous block's graph is kept alive until
            # this capture has ended -- a pool whose last graph was destroyed is released, and capturing into its stale handle trips
            # an allocator assert (measured) -- its tensors are dead, so the new capture reuses its memory.
            if self._graph_stream is None:
                self._graph_stream = torch.cuda.Stream(device)
            if self._graph_pool is None or self._last_graph is None:
                self._graph_pool = torch.cuda.graph_pool_handle()
            graph = torch.cuda.CUDAGraph()
            side = self._graph_stream
            side.wait_stream(torch.cuda.current_stream(device))
            with torch.cuda.stream(side):
                graph.capture_begin(pool=self._graph_pool)
                try:
                    body()                                              # recorded, not executed
                finally:
                    graph.capture_end()
            torch.cuda.current_stream(device).wait_stream(side)
        except RuntimeError as e:   # a failed capture (an op that synchronises, an allocation outside the pool): the same body runs
            graph = None            # eagerly instead -- still from the device tables (lr_table / index schedule), as iteration 0 did
            for a, fresh, acc in flags:
                a.wq_fresh = fresh
                for l, v in zip(a.layers, acc):
                    l._dw_accum[0] = v
            warnings.warn(f"hipGraph capture of the tuning iteration failed ({e!r}); running the iterations eagerly")
        finally:
            ops.profile_enable(prev_prof)
        self.last_hip_graph = graph is not None
        try:
            for _ in range(1, iters):
                if graph is not None:
                    graph.replay()
                else:
                    body()
        finally:
            optimizer.lr_override = None    # (also when a replay raises: the optimizer must not keep pointing at this block's tables)
        self._last_graph = graph            # (kept until the next block: replays may still be in flight)
        return iters

    def _loss_fwd_bwd(self, pred, ref, dpred, total_loss, n, num_elm, accum, tmask):
        """loss value into `total_loss` (+= loss/num_elm) and d(1000*loss)/dpred into `dpred`, one fused pass.
        reference: _get_loss + loss.item()/num_elm + _scale_loss_and_backward (sign_round/quantizer.py:127-158, :487-497)"""
        if accum:   # reduction="sum" and loss/num_elm in the reference (quantizer.py:436-452, :496)
            ops.mse_loss_fwd_bwd(pred, ref, dpred=dpred, loss_accum=total_loss, accum_scale=float(n) / float(num_elm),
                                 grad_scale=1000.0 * float(n), token_mask=tmask)
        else:
            ops.mse_loss_fwd_bwd(pred, ref, dpred=dpred, loss_accum=total_loss, accum_scale=1.0 / float(num_elm),
                                 grad_scale=1000.0, token_mask=tmask)

    def register_fp_input_forward_hooks(self, block) -> list:
        """Hooks that fire during the reference (fp-weight) forward of the block (quantization/base.py:61-67)."""
        return []

    def prepare_block(self, block) -> None:
        """Per-block scheme-dependent setup (the reference does this once per model in prepare_run)."""

    # ------------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def forward_all(self, block, inputs: torch.Tensor, input_others, batch_size: Optional[int] = None) -> torch.Tensor:
        """No-grad forward of every cached sample in minibatches -> [N, S, H] (composer.py steps 3 and 6)."""
        from .attention import reproducible_sdpa_forward

        with reproducible_sdpa_forward(bool(getattr(self.config, "reproducible_attention_forward", True)),
                                       exact=bool(getattr(self.config, "exact_attention", True))):
            return self._forward_all(block, inputs, input_others, batch_size)

    def _forward_all(self, block, inputs: torch.Tensor, input_others, batch_size: Optional[int] = None) -> torch.Tensor:
        bs = batch_size or self.config.batch_size
        fb = None
        if self.config.exact_rounding and self.config.amp and isinstance(input_others, dict) and not self.config.data_parallel:
            fb = self._build_exact_plain(block, inputs, input_others, min(bs, inputs.shape[0]))
            if fb is not None:
                outs = [fb.forward_nograd(inputs[b0:b0 + bs], self._others_for(min(bs, inputs.shape[0] - b0), input_others, consumer=fb))
                        for b0 in range(0, inputs.shape[0], bs)]
                return torch.cat(outs, dim=0)
        elif self.config.fused_block and self.config.amp and isinstance(input_others, dict):
            from .fused_block import build_fused_block_plain

            fb = build_fused_block_plain(block, input_others, self.config.amp_dtype, sdpa_ctx=self._sdpa_ctx)
            if fb is not None:
                fb.flash_fwd = bool(self.config.flash_attention)
                fb.flash_bwd = bool(self.config.flash_attention_bwd)
                key = ("plain", self._block_signature(block), self._others_signature(block, input_others))
                if key not in self._fused_verdict:
                    self._fused_verdict[key] = fb.agrees_with_module(lambda x, o: self.block_forward(block, x, o),
                                                                     inputs[:min(2, inputs.shape[0])], input_others)
                    self._report_fused_verdict(block, fb, self._fused_verdict[key])
                if not self._fused_verdict[key]:
                    fb = None
        outs = []
        for b0 in range(0, inputs.shape[0], bs):
            x = inputs[b0:b0 + bs]
            outs.append(fb.forward_nograd(x, input_others) if fb is not None else self.block_forward(block, x, input_others))
        return torch.cat(outs, dim=0)

    def calibrate_block(self, block, X: torch.Tensor, input_others, Xq: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Everything the composer does for a block BEFORE tuning it (composer.py:423-451): NVFP4 global scales (unified over
        q/k/v, gate/up, w1/w3), the scheme-dependent per-block setup, the reference forward with the fp weights under the
        calibration hooks (imatrix of the algorithm extension; act_max of statically activation-quantised layers -- taken from
        the fp-input forward, or, with quantised-input chaining, from one extra forward on the quantised input) and act_max for
        experts that saw no token.  -> fp_outputs [N, S, H].  Shared by `compress_block` and the block-sharded path."""
        update_block_global_scale_if_needed(block)      # composer.py:438-451 (NVFP4 only; no-op otherwise)
        self.prepare_block(block)
        need_q = bool(self.config.enable_quanted_input)
        handles = ([] if need_q else register_act_max_hooks(block)) + self.register_fp_input_forward_hooks(block)
        try:
            fp_out = self.forward_all(block, X, input_others)
        finally:
            for h in handles:
                h.remove()
        if need_q:
            handles = register_act_max_hooks(block)
            if handles:
                try:
                    self.forward_all(block, Xq if Xq is not None else X, input_others)
                finally:
                    for h in handles:
                        h.remove()
        set_amax_for_uncalibrated_experts(block)                  # composer.py:438-451
        return fp_out

    def compress_block(self, block, fp_inputs, input_others, q_inputs=None, block_ctx=None, input_ids=None):
        """reference: AlgorithmComposer.compress_block (composer.py:360-483):
        (3) reference forward with the fp weights, (4) quantize_block, (6) forward of the quantised block to produce
        the next block's quantised input.  -> (fp_outputs [N,S,H], q_outputs [N,S,H] or None, best_params)"""
        device = self.device
        with torch.cuda.device(device):
            X = stack_samples(fp_inputs, device)
            Xq = stack_samples(q_inputs, device) if (q_inputs is not None and self.config.enable_quanted_input) else None
            fp_out = self.calibrate_block(block, X, input_others, Xq)
            best = self.quantize_block(block, X, input_others, fp_out, Xq, block_ctx, input_ids=input_ids)
            q_out = None
            if self.config.enable_quanted_input:
                q_out = self.forward_all(block, Xq if Xq is not None else X, input_others)
        return fp_out, q_out, best


class SignRoundV2Quantizer(SignRoundQuantizer):
    """SignRound with the algorithm extension (reference: sign_roundv2/quantizer.py:321-428, `enable_alg_ext=True`):

    * symmetric int / mx_fp4 / nv_fp4 blocks are tuned through `SignRoundOptimizedWrapperLinear` (searched init scale
      weighted by the importance matrix, max_scale in (0, 2));
    * the importance matrix -- sum over calibration tokens of x^2 per input channel -- is collected by forward hooks
      while the block's reference outputs are computed (not for W-int4/A-int4);
    * when bits < 4 or act_bits <= 4 the loss drops the 0.1% largest |pred - ref| (`ar_outlier_mse_loss_fwd_bwd`).

    The double-quant (`*_dq`, GGUF) wrapper of the reference is outside the hot path."""

    def __init__(self, config: Optional[SignRoundConfig] = None, device: Union[str, torch.device] = "cuda", **kwargs):
        super().__init__(config, device, **kwargs)
        self._use_outlier_suppressed_loss = False
        self._optimized = False
        self._scheme = None

    @staticmethod
    def _block_scheme(block):
        for m in block.modules():
            if _quantizable(m) and check_to_quantized(m):
                return dict(bits=int(m.bits), sym=bool(m.sym), data_type=str(getattr(m, "data_type", "int")),
                            act_bits=int(getattr(m, "act_bits", 16)), act_data_type=str(getattr(m, "act_data_type", "")),
                            super_group_size=getattr(m, "super_group_size", None))
        return None

    def prepare_block(self, block) -> None:
        """reference: SignRoundV2Quantizer.prepare_run (sign_roundv2/quantizer.py:330-357), evaluated on the block's own
        scheme attributes (the reference reads the model-wide scheme)."""
        sch = self._scheme = self._block_scheme(block)
        self._optimized = False
        self._use_outlier_suppressed_loss = False
        self.wrapper_block = wrapper_block
        if sch is None:
            return
        dt = sch["data_type"]
        if dt.endswith("dq"):
            raise NotImplementedError("double-quant (GGUF *_dq) tuning is outside the MI355X hot path")
        if sch["sym"] and sch["super_group_size"] is None and (dt.startswith("int") or dt.startswith("mx") or dt.startswith("nv")):
            self._use_outlier_suppressed_loss = sch["act_bits"] <= 4 or sch["bits"] < 4
            self._optimized = True
            self.wrapper_block = functools.partial(wrapper_block, wrapper_cls=SignRoundOptimizedWrapperLinear)

    def _is_wint4aint4(self) -> bool:
        sch = self._scheme or {}
        adt, dt = sch.get("act_data_type", ""), sch.get("data_type", "")
        return (("int4" in adt or ("int" in adt and sch.get("act_bits") == 4))
                and ("int4" in dt or ("int" in dt and sch.get("bits") == 4)))

    def register_fp_input_forward_hooks(self, block) -> list:
        """imatrix hooks (sign_roundv2/quantizer.py:401-428): module.imatrix += sum_tokens x^2, fp32 [in_features]."""
        if self._scheme is None or self._is_wint4aint4():
            return []

        def collect_imatrix(module, inp, out):
            x = inp[0] if isinstance(inp, (tuple, list)) else inp
            sq = torch.sum(torch.pow(x.reshape(-1, x.shape[-1]).to(torch.float32), 2), dim=0).to(torch.float32)
            if not hasattr(module, "imatrix"):
                module.imatrix = sq
            else:
                module.imatrix += sq.to(module.imatrix.device)

        return [m.register_forward_hook(collect_imatrix) for m in block.modules()
                if _quantizable(m) and check_to_quantized(m)]

    def quantize_block(self, block, fp_inputs, input_others, fp_outputs, q_inputs=None, block_ctx=None, input_ids=None,
                       **kwargs) -> dict:
        # always derived from THIS block's scheme attributes (cheap, idempotent): quantize_block may be called directly, or --
        # block-sharded runs -- long after the calibration forward of the same block, with other blocks prepared in between
        self.prepare_block(block)
        try:
            return super().quantize_block(block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids, **kwargs)
        finally:
            self._scheme = None

    def _loss_fwd_bwd(self, pred, ref, dpred, total_loss, n, num_elm, accum, tmask):
        if not self._use_outlier_suppressed_loss:
            # the reference's V2 quantizer calls the base loss WITHOUT the valid-token mask (`super()._get_loss(pred_output,
            # ref_output, indices, mse_loss, device)`, sign_roundv2/quantizer.py:399): with the algorithm extension on and the
            # outlier-suppressed loss off (e.g. asymmetric schemes) every position enters the loss and its gradient, while the
            # divisor num_elm still counts the valid tokens -- mirrored, the tuned weights are held to the reference's
            return super()._loss_fwd_bwd(pred, ref, dpred, total_loss, n, num_elm, accum, None)
        # the outlier-suppressed loss is a mean regardless of the accumulation mode (sign_roundv2/quantizer.py:387-398)
        ops.outlier_mse_loss_fwd_bwd(pred, ref, dpred=dpred, loss_accum=total_loss, accum_scale=1.0 / float(num_elm),
                                     grad_scale=1000.0, token_mask=tmask)


def _name_of(block, module) -> str:
    for n, m in block.named_modules():
        if m is module:
            return n
    raise KeyError("wrapper not found in block")
