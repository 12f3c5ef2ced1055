"""Host logic of the plain "auto_gptq" / "auto_awq" checkpoint formats and of `layer_config` key resolution: known answers,
and -- where the reference tree is present -- the reference's own exporters run on CPU on the same tiny model."""
import json
import os
import sys
from types import SimpleNamespace

import pytest
import torch

from auto_round_amd.schemes import apply_scheme, expand_layer_config, layer_pattern_regex, resolve_scheme

REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")

# pattern -> regex; the first four are the reference's documented examples (utils/common.py:823-831)
KNOWN = {"model.embed_tokens": r".*model\.embed_tokens.*", "mlp.gate": r".*mlp\.gate.*", "mlp.gate$": r".*mlp\.gate$",
         "mlp.*gate": r".*mlp.*gate.*", "model.layers.[0-3].mlp": r".*model\.layers\.[0-3]\.mlp.*",
         "^model.layers.1.": r"^model\.layers\.1\..*", "q_proj|k_proj": r".*q_proj|k_proj.*", ".*down.*": r".*down.*",
         "a.b.*c.d": r".*a\.b.*c\.d.*", "(q|k)_proj": r".*(q|k)_proj.*"}


def _ref_on_path():
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)


def test_layer_pattern_regex_known_answers():
    for pat, rx in KNOWN.items():
        assert layer_pattern_regex(pat) == rx, pat
    with pytest.raises(Exception):
        layer_pattern_regex("layers.(0")            # an invalid pattern is an error, not a silent non-match


@needs_ref
def test_layer_pattern_regex_equals_the_reference():
    _ref_on_path()
    from auto_round.utils import to_standard_regex

    for pat in list(KNOWN) + ["down_proj", "x.y?z", "layers.1+.mlp", "model.layers.0.self_attn.q_proj", "experts.[0-9]+.w1$"]:
        assert layer_pattern_regex(pat) == to_standard_regex(pat), pat


def test_expand_layer_config_exact_names_and_patterns():
    names = ["model.layers.0.mlp.down_proj", "model.layers.1.mlp.down_proj", "model.layers.1.self_attn.q_proj",
             "model.layers.10.self_attn.q_proj"]
    got = expand_layer_config(names, {"model.layers.0.mlp.down_proj": {"bits": 16}, "layers.1.self_attn": {"bits": 8},
                                      "layers.[0-1].mlp": {"group_size": 64}, "nothing_like_this": {"bits": 2}})
    assert got == {"model.layers.0.mlp.down_proj": {"bits": 16, "group_size": 64},
                   "model.layers.1.mlp.down_proj": {"group_size": 64}, "model.layers.1.self_attn.q_proj": {"bits": 8}}
    assert expand_layer_config(names, None) == {}


def _shell(model, sym, layer_config, iters=1):
    """An AutoRound facade object with the state `quantize()` leaves behind, without a GPU (the config builders are host code)."""
    from auto_round_amd.autoround import AutoRound, _block_layer_config

    ar = AutoRound.__new__(AutoRound)
    ar.layer_config_in = layer_config
    ar.model, ar.scheme, ar.config = model, resolve_scheme("W4A16", group_size=32, sym=sym), SimpleNamespace(iters=iters)
    ar.block_names, ar.layer_config = [f"model.layers.{i}" for i in range(len(model.model.layers))], {}
    for n in ar.block_names:
        b = model.get_submodule(n)
        for ln, cfg in apply_scheme(b, ar.scheme, layer_config=_block_layer_config(layer_config, n, b)).items():
            ar.layer_config[f"{n}.{ln}"] = cfg
    return ar


def test_plain_format_configs_known_answers():
    from test_gpu_autoround import tiny_llama

    ar = _shell(tiny_llama(seed=3, vocab=64), True, None, iters=7)
    g = ar._gptq_quantization_config()
    assert g == {"bits": 4, "group_size": 32, "sym": True, "data_type": "int", "iters": 7, "autoround_version": g["autoround_version"],
                 "static_kv_granularity": "tensor", "static_attention_granularity": "tensor", "lm_head": False,
                 "provider": "auto-round", "quant_method": "gptq", "desc_act": False, "true_sequential": False, "damp_percent": 0.01}
    a = _shell(tiny_llama(seed=3, vocab=64), False, {"mlp.up_proj": {"bits": 16}})._awq_quantization_config()
    assert a["quant_method"] == "awq" and a["version"] == "gemm" and a["zero_point"] is True and a["to_quant_block_names"] == "model.layers"
    assert a["modules_to_not_convert"] == ["lm_head", "model.layers.0.mlp.up_proj", "model.layers.1.mlp.up_proj",
                                           "mlp.up_proj"]       # the pattern itself is listed too (export_to_awq/export.py:107-111)


@needs_ref
@pytest.mark.parametrize("fmt,sym,lc", [
    ("auto_gptq", True, {"model.layers.0.mlp.down_proj": {"bits": 16}, "model.layers.1.self_attn.q_proj": {"bits": 8}}),
    ("auto_gptq", True, {"model.layers.0.mlp.down_proj": {"bits": 16}, "k_proj": {"group_size": 64}, "layers.1.mlp": {"bits": 16}}),
    ("auto_awq", False, {"model.layers.0.mlp.down_proj": {"bits": 16}}),
    ("auto_awq", False, {"model.layers.0.mlp.down_proj": {"bits": 16}, "layers.1.mlp": {"bits": 16}}),
    ("auto_round", True, {"model.layers.0.mlp.down_proj": {"bits": 16}, "model.layers.1.self_attn.q_proj": {"bits": 8},
                          "k_proj": {"group_size": 64}})],
    ids=["gptq_names", "gptq_patterns", "awq_names", "awq_patterns", "auto_round_patterns"])
def test_plain_format_configs_equal_the_reference_export(fmt, sym, lc, tmp_path, monkeypatch):
    _ref_on_path()
    from auto_round import AutoRound

    from test_gpu_autoround import tiny_llama
    from test_pipeline_vs_reference import _Loader, _StubTokenizer

    monkeypatch.chdir(tmp_path)
    tokens = torch.randint(0, 64, (4, 16), generator=torch.Generator().manual_seed(1))
    ar = AutoRound(tiny_llama(seed=3, vocab=64), tokenizer=_StubTokenizer(), iters=1, nsamples=4, seqlen=16, dataset=_Loader(tokens),
                   device_map="cpu", batch_size=4, enable_torch_compile=False, scheme="W4A16", group_size=32, sym=sym,
                   layer_config={k: dict(v) for k, v in lc.items()})
    out = str(tmp_path / "ref")
    ar.quantize_and_save(out, format=fmt)
    sub = [d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))]
    ref_qc = json.load(open(os.path.join(out, sub[0], "config.json") if sub else os.path.join(out, "config.json")))["quantization_config"]
    mine = _shell(tiny_llama(seed=3, vocab=64), sym, lc)
    my_qc = {"auto_gptq": mine._gptq_quantization_config, "auto_awq": mine._awq_quantization_config,
             "auto_round": lambda: mine._quantization_config("auto_round:auto_gptq")}[fmt]()
    assert set(ref_qc) == set(my_qc), sorted(set(ref_qc) ^ set(my_qc))
    for k in ref_qc:
        if k == "modules_to_not_convert":           # built from a set in the reference: order is not part of the format
            assert sorted(ref_qc[k]) == sorted(my_qc[k])
        elif k != "autoround_version":
            assert ref_qc[k] == my_qc[k], k


def _fp4_shell(model, scheme_name, layer_config=None):
    from auto_round_amd.autoround import AutoRound, _block_layer_config

    ar = AutoRound.__new__(AutoRound)
    ar.layer_config_in = layer_config
    ar.model, ar.scheme, ar.config = model, resolve_scheme(scheme_name), SimpleNamespace(iters=1)
    ar.block_names, ar.layer_config = [f"model.layers.{i}" for i in range(len(model.model.layers))], {}
    for n in ar.block_names:
        b = model.get_submodule(n)
        for ln, cfg in apply_scheme(b, ar.scheme, layer_config=_block_layer_config(layer_config, n, b)).items():
            ar.layer_config[f"{n}.{ln}"] = cfg
    return ar


def test_llm_compressor_config_known_answers():
    from test_gpu_autoround import tiny_llama

    nv = _fp4_shell(tiny_llama(seed=3, vocab=64), "NVFP4", {"k_proj": {"bits": 16, "act_bits": 16}})._llmc_quantization_config()
    g = nv["config_groups"]["group_0"]
    assert nv["format"] == "nvfp4-pack-quantized" and nv["quant_method"] == "compressed-tensors" and nv["quantization_status"] == "compressed"
    assert g["weights"]["num_bits"] == 4 and g["weights"]["type"] == "float" and g["weights"]["strategy"] == "tensor_group"
    assert g["weights"]["group_size"] == 16 and g["weights"]["dynamic"] is False and g["input_activations"]["dynamic"] == "local"
    assert g["targets"] == ["Linear"] and nv["provider"] == "auto-round"
    assert nv["ignore"] == ["re:.*k_proj.*", "model.layers.0.self_attn.k_proj", "model.layers.1.self_attn.k_proj", "lm_head"]
    mx = _fp4_shell(tiny_llama(seed=3, vocab=64), "MXFP4")._llmc_quantization_config()
    g = mx["config_groups"]["group_0"]
    assert mx["format"] == "mxfp4-pack-quantized" and g["weights"]["group_size"] == 32 and g["weights"]["strategy"] == "group"
    assert g["input_activations"]["num_bits"] == 4 and g["input_activations"]["dynamic"] is True and mx["ignore"] == ["lm_head"]
    with pytest.raises(ValueError):
        _shell(tiny_llama(seed=3, vocab=64), True, None)._llmc_quantization_config()          # INT schemes: not this exporter


def test_llm_compressor_config_asks_compressed_tensors_when_it_is_importable(monkeypatch):
    """the literal is the fallback; with compressed-tensors importable the preset scheme's own `to_dict()` is what gets
    written (as in the reference, config.py:56-101).  The package is not installed here, so a stand-in with the three names the
    reference imports proves the route and what is passed to it; with the real package the literal is diffed against it."""
    import sys
    import types

    from test_gpu_autoround import tiny_llama

    seen = {}

    class _Cfg:
        def __init__(self, **kw):
            seen.update(kw)

        def to_dict(self):
            return {"config_groups": {"group_0": seen["config_groups"]["group_0"]}, "ignore": seen["ignore"], "format": self.format,
                    "quantization_status": seen["quantization_status"], "extra_field_of_a_newer_version": 1}

    fake = types.ModuleType("compressed_tensors.quantization")
    fake.QuantizationConfig = _Cfg
    fake.QuantizationStatus = SimpleNamespace(COMPRESSED="compressed")
    fake.preset_name_to_scheme = lambda name, targets: {"preset": name, "targets": targets}
    monkeypatch.setitem(sys.modules, "compressed_tensors", types.ModuleType("compressed_tensors"))
    monkeypatch.setitem(sys.modules, "compressed_tensors.quantization", fake)
    nv = _fp4_shell(tiny_llama(seed=3, vocab=64), "NVFP4")._llmc_quantization_config()
    assert nv["config_groups"]["group_0"] == {"preset": "NVFP4", "targets": ["Linear"]} and nv["extra_field_of_a_newer_version"] == 1
    assert nv["format"] == "nvfp4-pack-quantized" and nv["provider"] == "auto-round" and nv["ignore"] == ["lm_head"]
    assert seen["kv_cache_scheme"] is None


def test_llm_compressor_literal_matches_the_installed_compressed_tensors():
    ct = pytest.importorskip("compressed_tensors.quantization")
    from test_gpu_autoround import tiny_llama

    from auto_round_amd import autoround as ara

    for name in ("NVFP4", "MXFP4"):
        shell = _fp4_shell(tiny_llama(seed=3, vocab=64), name)
        live = shell._llmc_quantization_config()
        saved, ara._compressed_tensors_config = ara._compressed_tensors_config, lambda *a: None
        try:
            literal = shell._llmc_quantization_config()
        finally:
            ara._compressed_tensors_config = saved
        for side in ("weights", "input_activations"):
            lit, liv = literal["config_groups"]["group_0"][side], live["config_groups"]["group_0"][side]
            assert {k: liv.get(k) for k in lit} == lit, (name, side, ct.__name__, ara.LLMC_LITERAL_PINNED_TO)


@needs_ref
def test_llm_compressor_config_follows_the_reference_helpers_and_dict_layout():
    """compressed-tensors is absent, so the reference cannot build (or save) this config here; what it DOES hold without that
    package is pinned: the scheme / format names (`_get_scheme`, `_get_group_format`), the ignore list
    (`generate_ignore_regex_list`) and the dict layout it hard-codes for its NVFP4-E5M3 variant (config.py:103-139)."""
    _ref_on_path()
    from auto_round.export.export_to_llmcompressor.config import initialize_nvfp4_e5m3_quantization
    from auto_round.export.export_to_llmcompressor.utils import generate_ignore_regex_list

    from test_gpu_autoround import tiny_llama

    lc = {"k_proj": {"bits": 16, "act_bits": 16}, "model.layers.1.mlp.down_proj": {"bits": 16}}
    ar = _fp4_shell(tiny_llama(seed=3, vocab=64), "NVFP4", lc)
    mine = ar._llmc_quantization_config()
    ref_layout = initialize_nvfp4_e5m3_quantization(ignore=["lm_head"])
    assert set(mine) - {"provider"} == set(ref_layout)
    assert set(mine["config_groups"]["group_0"]) == set(ref_layout["config_groups"]["group_0"])
    for side in ("weights", "input_activations"):
        a, b = mine["config_groups"]["group_0"][side], ref_layout["config_groups"]["group_0"][side]
        assert set(a) == set(b) and all(a[k] == b[k] for k in a), side           # NVFP4 and its E5M3 variant share every quant arg
    for k in ("global_compression_ratio", "kv_cache_scheme", "quant_method", "quantization_status"):
        assert mine[k] == ref_layout[k], k
    # the format / scheme names: functions of export_to_fp.py that need nothing from compressed-tensors (the module imports it
    # lazily through config.py; _get_scheme / _get_group_format are plain string logic)
    import importlib

    try:
        fp = importlib.import_module("auto_round.export.export_to_llmcompressor.export_to_fp")
        assert fp._get_group_format(4, "nv_fp") == mine["format"] and fp._get_scheme(4, "nv_fp") == "NVFP4"
        assert fp._get_group_format(4, "mx_fp") == _fp4_shell(tiny_llama(seed=3, vocab=64), "MXFP4")._llmc_quantization_config()["format"]
    except ImportError:
        pass
    regex_config = {k: v for k, v in lc.items() if k not in ar.layer_config}
    ref_ignore = generate_ignore_regex_list(regex_config=regex_config, layer_config=ar.layer_config)
    assert mine["ignore"] == ref_ignore + ["lm_head"]
