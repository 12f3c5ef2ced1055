"""Quantisation scheme presets on the hot path (the reference's `auto_round/schemes.py:538-832`, same names and values)
and their application to a model's linears (what `apply_plan_to_model` does in the reference,
compressors/layer_config/resolver.py:482-497): each `nn.Linear` / `Conv1D` to be tuned receives `bits, group_size, sym,
data_type, scale_dtype, act_bits, act_data_type, act_group_size, act_sym, act_dynamic`."""
from __future__ import annotations

from typing import Dict, Iterable, Optional

import torch

_INT = dict(sym=True, data_type="int", act_bits=16, act_data_type="int", act_group_size=None, act_sym=None, act_dynamic=None)
PRESET_SCHEMES: Dict[str, dict] = {
    "W4A16": dict(_INT, bits=4, group_size=128),
    "W2A16": dict(_INT, bits=2, group_size=128),
    "W2A16G64": dict(_INT, bits=2, group_size=64),
    "W2A16G32": dict(_INT, bits=2, group_size=32),
    "W3A16": dict(_INT, bits=3, group_size=128),
    "W8A16": dict(_INT, bits=8, group_size=128),
    # per-channel weights + dynamic per-token symmetric activations
    "INT8": dict(bits=8, group_size=-1, sym=True, data_type="int", act_bits=8, act_data_type="int", act_group_size=-1,
                 act_sym=True, act_dynamic=True),
    "INT8_W8A8": dict(bits=8, group_size=-1, sym=True, data_type="int", act_bits=8, act_data_type="int", act_group_size=-1,
                      act_sym=True, act_dynamic=True),
    "INT4": dict(bits=4, group_size=-1, sym=True, data_type="int", act_bits=4, act_data_type="int", act_group_size=-1,
                 act_sym=True, act_dynamic=True),
    "MXFP4": dict(bits=4, group_size=32, sym=True, data_type="mx_fp", act_bits=4, act_data_type="mx_fp", act_group_size=32,
                  act_sym=True, act_dynamic=True),
    "NVFP4": dict(bits=4, group_size=16, sym=True, data_type="nv_fp", act_bits=4, act_data_type="nv_fp4_with_static_gs",
                  act_group_size=16, act_sym=True, act_dynamic=True),
}
SCHEME_KEYS = ("bits", "group_size", "sym", "data_type", "act_bits", "act_data_type", "act_group_size", "act_sym", "act_dynamic")


def resolve_scheme(scheme="W4A16", **overrides) -> dict:
    """Preset name (or a dict with the same keys) + explicit overrides (`bits=`, `group_size=`, `sym=`, ...; None = keep);
    reference: PRESET_SCHEMES and scheme resolution, auto_round/schemes.py:538-832."""
    if isinstance(scheme, str):
        if scheme.upper() not in PRESET_SCHEMES:
            raise ValueError(f"scheme {scheme!r}: the MI355X path implements {sorted(PRESET_SCHEMES)}")
        cfg = dict(PRESET_SCHEMES[scheme.upper()])
    else:
        cfg = dict(scheme)
    for k, v in overrides.items():
        if v is not None:
            if k not in SCHEME_KEYS:
                raise KeyError(k)
            cfg[k] = v
    return cfg


_REGEX_TOKENS = (".*", "^", "$", "|", "(", ")", "[", "]", "?", "+")


def layer_pattern_regex(pattern: str) -> str:
    """How a `layer_config` key that is not an exact layer name is read (reference: utils/common.py:813-867
    `to_standard_regex`): a plain string matches as a substring (escaped, wrapped in `.*`); a string that already contains
    regex tokens keeps them, has its bare dots escaped and is opened on the sides it does not anchor itself."""
    import re

    if not any(t in pattern for t in _REGEX_TOKENS):
        return f".*{re.escape(pattern)}.*"
    rx = re.sub(r"\.(\*?)", lambda m: ".*" if m.group(1) else "\\.", pattern)
    if not rx.startswith(("^", ".*")):
        rx = ".*" + rx
    if not rx.endswith(("$", ".*")):
        rx += ".*"
    re.compile(rx)
    return rx


def expand_layer_config(layer_names: Iterable[str], layer_config: Optional[Dict[str, dict]]) -> Dict[str, dict]:
    """-> {exact layer name: overrides}.  Exact names are taken as they are; every other key is a pattern searched in the
    layer names (reference: compressors/layer_config/resolver.py:289-322)."""
    import re

    names = list(layer_names)
    out: Dict[str, dict] = {}
    for key, over in (layer_config or {}).items():
        if key in names:
            out.setdefault(key, {}).update(over)
            continue
        rx = re.compile(layer_pattern_regex(key))
        for n in names:
            if rx.search(n):
                out.setdefault(n, {}).update(over)
    return out


def is_quantizable(module) -> bool:
    try:
        from transformers.pytorch_utils import Conv1D
    except Exception:  # pragma: no cover
        Conv1D = ()
    return isinstance(module, torch.nn.Linear) or (bool(Conv1D) and isinstance(module, Conv1D))


def apply_scheme(root: torch.nn.Module, scheme: dict, skip: Iterable[str] = ("mlp.gate", "router", "lm_head"),
                 layer_config: Optional[Dict[str, dict]] = None, scale_dtype=torch.float16) -> Dict[str, dict]:
    """Attach the scheme attributes to every quantisable linear under `root` and return `{layer name: config}`.
    Layers whose name ends with an entry of `skip` (MoE router gates, lm_head) and layers whose shapes the packers cannot
    take (in/out features not divisible by 32, reference: check_to_quantized / export shape checks) stay 16-bit.
    `layer_config` overrides single layers by (suffix of the) name, like the reference's `layer_config` argument."""
    out = {}
    for name, m in root.named_modules():
        if not is_quantizable(m):
            continue
        cfg = dict(scheme)
        for pat, over in (layer_config or {}).items():
            if name == pat or name.endswith("." + pat):
                cfg.update(over)
        w = m.weight
        if any(name == s or name.endswith("." + s) or name.endswith(s) for s in skip) or w.shape[0] % 32 or w.shape[1] % 32:
            cfg["bits"], cfg["act_bits"] = 16, 16
        for k in SCHEME_KEYS:
            setattr(m, k, cfg.get(k))
        if m.act_bits is None:
            m.act_bits = 16
        m.scale_dtype = scale_dtype
        out[name] = {k: cfg.get(k) for k in SCHEME_KEYS}
    return out
