"""Format parity with the consumer side: checkpoints WRITTEN by this repository's GPU path (tests/golden/tiny_ckpt_*, made
on an MI355X by tools/make_tiny_ckpt.py through `AutoRound(...).quantize_and_save()`) are LOADED by the reference's own
inference stack -- transformers' auto-round quantizer -> auto_round.inference.convert_hf_model -> the reference's torch
QuantLinear (`auto_round_extension/torch/qlinear_torch[_zp].py`) -- and must reproduce the tuned model's logits.
Needs /root/reference (build container); the fixtures themselves are also decoded without it."""
import glob
import json
import os
import sys

import numpy as np
import pytest
import torch

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
REF = "/root/reference"
CKPTS = sorted(glob.glob(os.path.join(GOLDEN, "tiny_ckpt_*")))


@pytest.mark.parametrize("ck", CKPTS, ids=[os.path.basename(c)[10:] for c in CKPTS])
def test_fixture_checkpoint_layout(ck):
    from safetensors import safe_open

    qc = json.load(open(os.path.join(ck, "config.json")))["quantization_config"]
    if ck.endswith("_fmt_llmc"):        # compressed-tensors config + the fp4 tensor layout (export_to_llmcompressor/export_to_fp.py)
        nv = "nvfp4" in os.path.basename(ck)
        g0 = qc["config_groups"]["group_0"]
        assert qc["format"] == ("nvfp4-pack-quantized" if nv else "mxfp4-pack-quantized") and qc["quant_method"] == "compressed-tensors"
        assert g0["weights"]["num_bits"] == 4 and g0["input_activations"]["num_bits"] == 4 and qc["ignore"] == ["lm_head"]
        with safe_open(os.path.join(ck, "model.safetensors"), "pt") as f:     # what the reference's own tests check of this format
            keys = set(f.keys())                                              # (test/unit/test_cpu/quantization/test_mxfp_nvfp.py:150-170, :260-275)
            wp = f.get_tensor("model.layers.0.mlp.down_proj.weight_packed")
            ws = f.get_tensor("model.layers.0.mlp.down_proj.weight_scale")
        assert wp.dtype == torch.uint8 and tuple(wp.shape) == (128, 128) and ws.shape[0] == 128
        assert ws.dtype == (torch.float8_e4m3fn if nv else torch.uint8)
        assert ("model.layers.0.mlp.down_proj.weight_global_scale" in keys) == nv
        assert ("model.layers.0.mlp.down_proj.input_global_scale" in keys) == nv
        return
    bits, gs = qc["bits"], qc["group_size"]
    if "gpt2" in os.path.basename(ck):   # Conv1D layers are packed like linears of shape [out, in] (export.py:200-205)
        from test_gpu_autoround import unpack_w4_gptq

        z = np.load(os.path.join(ck, "expected.npz"))
        with safe_open(os.path.join(ck, "model.safetensors"), "pt") as f:
            keys = set(f.keys())
            for n in z["names"]:
                t = {k: f.get_tensor(f"{z['prefix']}{n}.{k}") for k in ("qweight", "qzeros", "scales")}
                W = unpack_w4_gptq(t["qweight"], t["qzeros"], t["scales"], gs)                 # [out, in]
                stored = torch.from_numpy(z["W_" + str(n).replace(".", "_")]).view(torch.bfloat16)      # Conv1D: [in, out]
                assert tuple(t["qweight"].shape) == (stored.shape[0] // 8, stored.shape[1])
                assert torch.equal(W.to(torch.bfloat16), stored.t()), n
        assert f"{z['prefix']}attn.c_attn.weight" not in keys and f"{z['prefix']}attn.c_attn.bias" in keys
        return
    if "_fmt_" in ck:                   # the plain auto_gptq / auto_awq layouts
        from test_gpu_autoround import unpack_w4_awq, unpack_w4_gptq

        awq = ck.endswith("awq")
        assert qc["quant_method"] == ("awq" if awq else "gptq") and qc["provider"] == "auto-round" and "packing_format" not in qc
        z = np.load(os.path.join(ck, "expected.npz"))
        with safe_open(os.path.join(ck, "model.safetensors"), "pt") as f:
            keys = set(f.keys())
            for n, key in (("self_attn.q_proj", "W_self_attn_q_proj"), ("mlp.down_proj", "W_mlp_down_proj")):
                t = {k: f.get_tensor(f"model.layers.0.{n}.{k}") for k in ("qweight", "qzeros", "scales")}
                assert t["qweight"].dtype == torch.int32 and t["scales"].dtype == torch.float16
                W = (unpack_w4_awq if awq else unpack_w4_gptq)(t["qweight"], t["qzeros"], t["scales"], gs)
                assert torch.equal(W.to(torch.bfloat16), torch.from_numpy(z[key]).view(torch.bfloat16)), n   # decodes to the tuned weights
        assert ("model.layers.0.mlp.down_proj.g_idx" in keys) == (not awq) and "lm_head.weight" in keys
        return
    assert qc["quant_method"] == "auto-round" and qc["packing_format"].startswith("auto_round")
    if qc["data_type"] != "int":        # MXFP4 / NVFP4: llm_compressor tensor layout
        assert qc["packing_format"] == "auto_round:llm_compressor"
        with safe_open(os.path.join(ck, "model.safetensors"), "pt") as f:
            keys = set(f.keys())
            wp = f.get_tensor("model.layers.0.mlp.down_proj.weight_packed")
            ws = f.get_tensor("model.layers.0.mlp.down_proj.weight_scale")
        assert wp.dtype == torch.uint8 and tuple(wp.shape) == (128, 256 // 2) and tuple(ws.shape) == (128, 256 // gs)
        assert ws.dtype == (torch.uint8 if qc["data_type"] == "mx_fp" else torch.float8_e4m3fn)
        nv = qc["data_type"] == "nv_fp"
        assert ("model.layers.0.mlp.down_proj.weight_global_scale" in keys) == nv
        assert ("model.layers.0.mlp.down_proj.input_global_scale" in keys) == nv
        return
    with safe_open(os.path.join(ck, "model.safetensors"), "pt") as f:
        keys = set(f.keys())
        qw = f.get_tensor("model.layers.0.mlp.down_proj.qweight")
        qz = f.get_tensor("model.layers.0.mlp.down_proj.qzeros")
        sc = f.get_tensor("model.layers.0.mlp.down_proj.scales")
    in_f, out_f = 256, 128
    assert qw.dtype == torch.int32 and tuple(qw.shape) == (in_f // 32 * bits, out_f)
    assert qz.dtype == torch.int32 and tuple(qz.shape) == (in_f // gs, out_f // 32 * bits)
    assert sc.dtype == torch.float16 and tuple(sc.shape) == (in_f // gs, out_f)
    assert "model.layers.0.mlp.down_proj.weight" not in keys and "lm_head.weight" in keys and "model.norm.weight" in keys


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("ck", CKPTS, ids=[os.path.basename(c)[10:] for c in CKPTS])
def test_reference_inference_stack_loads_our_checkpoint(ck):
    if ck.endswith("_fmt_llmc"):
        pytest.skip("llm_compressor checkpoints are loaded by vLLM / compressed-tensors, neither of which is installed; layout and "
                    "config are pinned by the fixture-layout and config tests")
    if ck.endswith("fmt_awq"):
        pytest.skip("the reference has no torch-only backend for the AWQ layout (inference/backend.py: awq backends need "
                    "gptqmodel / autoawq / auto-round-lib); the layout is pinned by the decode and structure tests")
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    import auto_round  # noqa: F401  (registers its loader with transformers)
    from auto_round.inference import backend as B
    from transformers import AutoModelForCausalLM, AutoRoundConfig

    saved = {}
    for k, info in B.BackendInfos.items():     # the reference is on sys.path, not pip-installed: its own version pin cannot hold
        if "torch" in k:
            saved[k] = info.requirements
            info.requirements = [r for r in (info.requirements or []) if not r.startswith("auto-round")]
    try:
        m = AutoModelForCausalLM.from_pretrained(ck, device_map="cpu", torch_dtype=torch.bfloat16,
                                                 quantization_config=AutoRoundConfig(backend="torch"))
    finally:
        for k, r in saved.items():
            B.BackendInfos[k].requirements = r
    z = np.load(os.path.join(ck, "expected.npz"))
    if "gpt2" in os.path.basename(ck):
        ql = m.transformer.h[0].attn.c_attn
        with torch.no_grad():
            logits = m(input_ids=torch.from_numpy(z["tokens"])).logits.float().numpy()
        assert type(ql).__module__.startswith("auto_round_extension.torch.qlinear_torch"), type(ql)
        assert np.abs(logits - z["logits"]).max() <= 0.05 * np.abs(z["logits"]).mean()
        return
    ql = m.model.layers[0].self_attn.q_proj
    fp4 = "fp4" in os.path.basename(ck)
    with torch.no_grad():
        logits = m(input_ids=torch.from_numpy(z["tokens"])).logits.float().numpy()
    scale = np.abs(z["logits"]).mean()
    if not fp4:
        assert type(ql).__module__.startswith("auto_round_extension.torch.qlinear_torch"), type(ql)
        # same integer weights and scales, bf16 GEMMs on a different device / summation order
        assert np.abs(logits - z["logits"]).max() <= 0.05 * scale
        return
    # MXFP4 / NVFP4: the reference's own QuantLinear decodes our nibbles + scales to EXACTLY the tuned weights ...
    assert type(ql).__module__.startswith("auto_round.experimental.qmodules"), type(ql)
    for n, key in (("self_attn.q_proj", "W_self_attn_q_proj"), ("mlp.down_proj", "W_mlp_down_proj")):
        w = m.get_submodule(f"model.layers.0.{n}").dequant_weight_online().to(torch.bfloat16)
        assert torch.equal(w, torch.from_numpy(z[key]).view(torch.bfloat16)), n
    # ... while 4-bit ACTIVATION rounding makes the logits sensitive to the host's arithmetic (fp32 CPU qdq here vs bf16 GPU)
    assert np.abs(logits - z["logits"]).mean() <= 0.2 * scale


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
@pytest.mark.parametrize("tag,kw", [("w4g32_sym", dict(scheme="W4A16", group_size=32)), ("nvfp4", dict(scheme="NVFP4")),
                                    ("w4g32_sym_fmt_gptq", dict(scheme="W4A16", group_size=32, format="auto_gptq")),
                                    ("w4g32_asym_fmt_awq", dict(scheme="W4A16", group_size=32, sym=False, format="auto_awq"))])
def test_checkpoint_structure_equals_a_reference_saved_checkpoint(tag, kw, tmp_path, monkeypatch):
    """The reference quantises and SAVES the same architecture on CPU (format "auto_round", "auto_gptq", "auto_awq"); our
    GPU-written fixture must have exactly the same tensor names, dtypes and shapes and the same quantization_config keys /
    values (versions and the run's iteration count aside)."""
    from safetensors import safe_open

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round import AutoRound

    from test_gpu_autoround import tiny_llama
    from test_pipeline_vs_reference import _Loader, _StubTokenizer

    monkeypatch.chdir(tmp_path)
    kw = dict(kw)
    fmt = kw.pop("format", "auto_round")
    tokens = torch.randint(0, 64, (4, 16), generator=torch.Generator().manual_seed(1))
    ar = AutoRound(tiny_llama(seed=3, vocab=64), tokenizer=_StubTokenizer(), iters=1, nsamples=4, seqlen=16, dataset=_Loader(tokens),
                   device_map="cpu", batch_size=4, enable_torch_compile=False, **kw)
    out = str(tmp_path / "ref")
    ar.quantize_and_save(out, format=fmt)
    sub = [d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))]
    out = os.path.join(out, sub[0]) if sub else out

    def load(d):
        t = {}
        for f in os.listdir(d):
            if f.endswith(".safetensors"):
                with safe_open(os.path.join(d, f), "pt") as sf:
                    for k in sf.keys():
                        x = sf.get_tensor(k)
                        t[k] = (x.dtype, tuple(x.shape))
        return t

    mine_dir = os.path.join(GOLDEN, f"tiny_ckpt_{tag}")
    ref_t, my_t = load(out), load(mine_dir)
    assert ref_t == my_t                                   # names, dtypes, shapes
    ref_qc = json.load(open(os.path.join(out, "config.json")))["quantization_config"]
    my_qc = json.load(open(os.path.join(mine_dir, "config.json")))["quantization_config"]
    assert set(ref_qc) == set(my_qc), (sorted(set(ref_qc) ^ set(my_qc)))
    for k in ref_qc:
        if k not in ("autoround_version", "iters"):
            assert ref_qc[k] == my_qc[k], k
    assert os.path.exists(os.path.join(out, "quantization_config.json")) and os.path.exists(os.path.join(mine_dir, "quantization_config.json"))
    if fmt != "auto_round":       # both plain exporters advertise fp16 to their consumers
        assert json.load(open(os.path.join(out, "config.json")))["torch_dtype"] == "float16"
        assert json.load(open(os.path.join(mine_dir, "config.json")))["torch_dtype"] == "float16"
