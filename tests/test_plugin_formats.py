"""P4 through the reference's own registry: `auto_round_amd.plugin.register_formats()` re-registers the
reference's "auto_round*", "auto_gptq" and "auto_awq" format names (`OutputFormat.register`, auto_round/export/formats/base.py:119-129)
with subclasses whose `pack_layer` runs this package's packers, so that the reference's own `quantize_and_save()` packs on the MI355X
with no edit of the reference.  CPU, live reference (build container only): the reference's front door runs twice on the same tiny
model -- once with its own format classes, once with the registered ones -- and every tensor of the two checkpoints must be equal
byte for byte, config.json's quantization_config too.  The HIP kernels cannot run here, so `ops.pack_int / pack_awq / pack_fp4` are
stood in for by the C oracle's packers behind the SAME signatures (the kernels themselves are held to the reference's packed words by
tests/test_gpu_kernels.py::test_pack_* on the GPU box); what this test pins is everything between the reference's registry and those
three calls: format resolution, which packer a backend string selects, buffer names / shapes / dtypes, bias handling, the module swap,
what falls through to the reference's own pack_layer."""
import copy
import json
import os
import sys

import numpy as np
import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")


def _oracle_packers(monkeypatch):
    """ops.pack_* with the oracle behind them, CPU tensors in and out, same signatures and return types as auto_round_amd/ops.py"""
    import auto_round_amd.export as export
    import auto_round_amd.ops as ops
    import auto_round_amd.plugin as plugin
    from oracle import oracle as orc

    calls = {"pack_int": 0, "pack_awq": 0, "pack_fp4": 0}

    def zp_arg(zp):
        return zp.detach().float().cpu().numpy() if isinstance(zp, torch.Tensor) else float(zp)

    def pack_int(Wq2d, scale2d, zp, *, gs, bits, zp_off=1):
        calls["pack_int"] += 1
        out_f, in_f = Wq2d.shape
        qw, qz, st = orc.pack_int(orc.to_bits(Wq2d).reshape(-1), orc.to_bits(scale2d).reshape(-1), zp_arg(zp), out_f, in_f, gs, bits,
                                  orc.dt_code(Wq2d.dtype), orc.dt_code(scale2d.dtype), zp_off=zp_off)
        return torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(st.view(np.int16)).view(torch.float16)

    def pack_awq(Wq2d, scale2d, zp, *, gs):
        calls["pack_awq"] += 1
        out_f, in_f = Wq2d.shape
        qw, qz, st = orc.pack_awq(orc.to_bits(Wq2d).reshape(-1), orc.to_bits(scale2d).reshape(-1), zp_arg(zp), out_f, in_f, gs,
                                  orc.dt_code(Wq2d.dtype), orc.dt_code(scale2d.dtype))
        return torch.from_numpy(qw), torch.from_numpy(qz), torch.from_numpy(st.view(np.int16)).view(torch.float16)

    def pack_fp4(Wq2d, scale, *, mode, gs, global_scale=None):
        calls["pack_fp4"] += 1
        out_f, in_f = Wq2d.shape
        sc = orc.to_bits(scale).reshape(-1) if mode == 0 else scale.detach().float().cpu().numpy().reshape(-1)
        packed, sb = orc.pack_fp4(orc.to_bits(Wq2d).reshape(-1), sc, out_f, in_f, gs, mode, orc.dt_code(Wq2d.dtype),
                                  global_scale=1.0 if global_scale is None else float(global_scale))
        return torch.from_numpy(packed), torch.from_numpy(sb)

    monkeypatch.setattr(ops, "pack_int", pack_int)
    monkeypatch.setattr(ops, "pack_awq", pack_awq)
    monkeypatch.setattr(ops, "pack_fp4", pack_fp4)
    monkeypatch.setattr(export, "_pack_device", lambda device, weight: torch.device("cpu"))
    monkeypatch.setattr(plugin, "_hip_available", lambda: True)
    return calls


def _load_checkpoint(folder):
    from safetensors.torch import load_file

    tensors = {}
    # (the reference appends a "<model>-w4g32"-style directory of its own to the path it is given)
    folder = next(d for d, _, files in os.walk(folder) if "config.json" in files)
    for f in sorted(os.listdir(folder)):
        if f.endswith(".safetensors"):
            tensors.update(load_file(os.path.join(folder, f)))
    with open(os.path.join(folder, "config.json")) as f:
        cfg = json.load(f)
    return tensors, cfg.get("quantization_config")


CASES = [
    ("w4g32_sym_auto_round", dict(scheme="W4A16", group_size=32), "auto_round", "pack_int"),          # -> auto_round:auto_gptq (zp - 1)
    ("w4g32_asym_auto_round", dict(scheme="W4A16", group_size=32, sym=False), "auto_round", "pack_awq"),   # -> auto_round:auto_awq
    ("w2g32_asym_auto_round", dict(scheme="W2A16G32", sym=False), "auto_round", "pack_int"),          # -> plain auto_round (zp)
    ("w3g32_sym_auto_round", dict(scheme="W3A16", group_size=32), "auto_round", "pack_int"),
    ("w4g32_sym_auto_gptq", dict(scheme="W4A16", group_size=32), "auto_gptq", "pack_int"),            # plain GPTQ layout (+ g_idx)
    ("mxfp4_auto_round", dict(scheme="MXFP4"), "auto_round", "pack_fp4"),
    ("nvfp4_auto_round", dict(scheme="NVFP4"), "auto_round", "pack_fp4"),
    ("w4a8_auto_round", dict(scheme="W4A16", group_size=32, act_bits=8), "auto_round", None),         # W4A8 container: falls through
]


@pytest.mark.parametrize("name,kw,fmt,packer", CASES, ids=[c[0] for c in CASES])
def test_registered_formats_write_the_reference_checkpoint_byte_for_byte(name, kw, fmt, packer, tmp_path, monkeypatch):
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round import AutoRound
    from auto_round.export.formats.base import OutputFormat

    import auto_round_amd.plugin as plugin
    from test_pipeline_vs_reference import _Loader, _StubTokenizer, _tiny

    monkeypatch.chdir(tmp_path)
    base = _tiny()
    tokens = torch.randint(0, 64, (8, 16), generator=torch.Generator().manual_seed(1))
    common = dict(tokenizer=_StubTokenizer(), nsamples=8, seqlen=16, dataset=_Loader(tokens), device_map="cpu", batch_size=4,
                  enable_torch_compile=False, iters=2, **kw)
    registry_before = dict(OutputFormat._format_list)
    try:
        AutoRound(copy.deepcopy(base), **common).quantize_and_save(str(tmp_path / "ref"), format=fmt)      # the reference's own packers
        calls = _oracle_packers(monkeypatch)
        made = plugin.register_formats()
        assert made or all(getattr(c, "mi355x_hip_packers", False) for n, c in OutputFormat._format_list.items()
                           if n in ("auto_round", "auto_gptq", "auto_awq"))
        for n in ("auto_round", "auto_round:auto_gptq", "auto_round:auto_awq", "auto_gptq", "auto_awq"):
            assert getattr(OutputFormat._format_list[n], "mi355x_hip_packers", False), n
        plugin.HIP_PACK_STATS.update(hip=0, reference=0)
        AutoRound(copy.deepcopy(base), **common).quantize_and_save(str(tmp_path / "hip"), format=fmt)      # same front door, registered formats
    finally:
        OutputFormat._format_list.clear()
        OutputFormat._format_list.update(registry_before)
    if packer is None:      # nothing for the HIP packers here: every layer must have gone to the reference's own pack_layer
        assert plugin.HIP_PACK_STATS["hip"] == 0 and sum(calls.values()) == 0, (plugin.HIP_PACK_STATS, calls)
    else:                   # 2 blocks x 7 linears, each through the expected packer and none through the reference's
        assert calls[packer] == 14 and sum(calls.values()) == 14 and plugin.HIP_PACK_STATS == {"hip": 14, "reference": 0}, (calls, plugin.HIP_PACK_STATS)
    t_ref, q_ref = _load_checkpoint(tmp_path / "ref")
    t_hip, q_hip = _load_checkpoint(tmp_path / "hip")
    assert sorted(t_ref) == sorted(t_hip), sorted(set(t_ref) ^ set(t_hip))[:10]
    for k, a in t_ref.items():
        b = t_hip[k]
        assert a.dtype == b.dtype and a.shape == b.shape, (k, a.dtype, b.dtype, a.shape, b.shape)
        assert a.contiguous().view(torch.uint8).numpy().tobytes() == b.contiguous().view(torch.uint8).numpy().tobytes(), k
    assert q_ref == q_hip
    assert packer is None or any(k.endswith((".qweight", ".weight_packed")) for k in t_ref)


def test_register_is_idempotent_and_leaves_other_formats_alone(monkeypatch):
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.export.formats.base import OutputFormat

    import auto_round_amd.plugin as plugin

    before = dict(OutputFormat._format_list)
    try:
        made = plugin.register_formats()
        again = plugin.register_formats()
        assert again == {} and set(made) >= {"auto_round", "auto_round:auto_gptq", "auto_round:auto_awq", "auto_gptq", "auto_awq"}
        for n, c in OutputFormat._format_list.items():
            if n in made:
                assert issubclass(c, before[n]) and c.support_schemes == before[n].support_schemes and c.format_name == before[n].format_name
            else:
                assert c is before[n], n          # gguf, fake, fp8, mlx, llm_compressor ...: untouched
    finally:
        OutputFormat._format_list.clear()
        OutputFormat._format_list.update(before)
