"""Whole-pipeline pin of the block-by-block driver flow (fp forward -> tune -> quantised-output forward -> chaining, token
mask, seeding, best-parameter selection) against the REAL reference front door on CPU: `AutoRound(...).quantize()` of the
reference vs the same flow written with oracle/torch_ref (what tests/e2e_engine_compare.py and the GPU driver
`auto_round_amd.model_tuner.tune_blocks` implement), on a tiny random Llama.  Same torch ops on the same device, so the tuned
weights must be IDENTICAL.  The reference's input cache hands the blocks the boolean causal mask cast to bf16
(calibration/inputs.py:100-107); the restatement is given the same tensor."""
import copy
import os
import sys

import pytest
import torch

from pipeline_flow import run_flow

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")


class _StubTokenizer:
    pad_token_id = None
    pad_token = None

    def save_pretrained(self, *a, **k):
        pass


class _Loader:
    batch_size = 1

    def __init__(self, tokens):
        self.tokens = tokens

    def __iter__(self):
        for r in self.tokens:
            yield r.reshape(1, -1)


def _tiny():
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(0)
    cfg = LlamaConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2,
                      vocab_size=64, max_position_embeddings=32, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    return LlamaForCausalLM(cfg).to(torch.bfloat16).eval()


def _tiny_opt():
    """The north-star's model family (OPT: learned positions, biases on every projection, ReLU MLP, model.decoder.layers)."""
    from transformers import OPTConfig, OPTForCausalLM

    torch.manual_seed(0)
    cfg = OPTConfig(hidden_size=64, ffn_dim=128, num_attention_heads=4, num_hidden_layers=2, vocab_size=64,
                    max_position_embeddings=32, word_embed_proj_dim=64)
    cfg._attn_implementation = "sdpa"
    return OPTForCausalLM(cfg).to(torch.bfloat16).eval()


def _tiny_gpt2():
    """GPT-2: Conv1D projections (weights stored [in, out]), fused qkv, transformer.h."""
    from transformers import GPT2Config, GPT2LMHeadModel

    torch.manual_seed(0)
    cfg = GPT2Config(n_embd=64, n_head=4, n_layer=2, n_inner=128, vocab_size=64, n_positions=32)
    cfg._attn_implementation = "sdpa"
    return GPT2LMHeadModel(cfg).to(torch.bfloat16).eval()


def _tiny_hf(kind):
    """Other decoder families through the same flow: Qwen2 (q/k/v biases), Qwen3 (q/k norms), Qwen3-MoE (fused 3-D experts)."""
    import transformers as T

    torch.manual_seed(0)
    common = dict(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2,
                  vocab_size=64, max_position_embeddings=32, tie_word_embeddings=False)
    if kind == "qwen2":
        cfg, cls = T.Qwen2Config(**common), T.Qwen2ForCausalLM
    elif kind == "qwen3":
        cfg, cls = T.Qwen3Config(head_dim=16, **common), T.Qwen3ForCausalLM
    else:
        cfg = T.Qwen3MoeConfig(head_dim=16, moe_intermediate_size=64, num_experts=4, num_experts_per_tok=2, decoder_sparse_step=1,
                               mlp_only_layers=[], **common)
        cls = T.Qwen3MoeForCausalLM
    cfg._attn_implementation = "sdpa"
    return cls(cfg).to(torch.bfloat16).eval()


def _layers(model):
    if hasattr(model, "transformer"):
        return model.transformer.h
    return model.model.decoder.layers if hasattr(model.model, "decoder") else model.model.layers


def _tiny_moe(experts=4, top_k=2):
    from transformers import MixtralConfig, MixtralForCausalLM

    torch.manual_seed(0)
    cfg = MixtralConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=2,
                        vocab_size=64, max_position_embeddings=32, num_local_experts=experts, num_experts_per_tok=top_k,
                        tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    return MixtralForCausalLM(cfg).to(torch.bfloat16).eval()


@pytest.mark.parametrize("kw", [dict(scheme="W4A16", group_size=32), dict(scheme="W2A16G32", sym=False), dict(scheme="MXFP4"),
                                dict(scheme="W2A16G32", enable_alg_ext=True), dict(scheme="NVFP4", enable_alg_ext=True),
                                dict(scheme="W2A16G32", sym=False, enable_alg_ext=True),
                                dict(scheme="W4A16", group_size=32, moe=True), dict(scheme="NVFP4", moe=(24, 1)),
                                dict(scheme="W4A16", group_size=32, arch="opt"), dict(scheme="W2A16G32", sym=False, arch="opt"),
                                dict(scheme="W4A16", group_size=32, arch="gpt2"), dict(scheme="W4A16", group_size=32, arch="qwen2"),
                                dict(scheme="W4A16", group_size=32, arch="qwen3"), dict(scheme="W4A16", group_size=32, arch="qwen3_moe", moe_arch=True),
                                dict(scheme="W4A16", group_size=32, act_bits=8), dict(scheme="INT8"), dict(scheme="W3A16", group_size=32),
                                dict(scheme="W8A16", group_size=32),
                                dict(scheme="W4A16", group_size=32, gradient_accumulate_steps=2),
                                dict(scheme="W4A16", group_size=32, not_use_best_mse=True),
                                dict(scheme="W4A16", group_size=32, enable_minmax_tuning=False),
                                dict(scheme="W4A16", group_size=32, enable_quanted_input=False),
                                dict(scheme="W2A16G32", lr=5e-3, minmax_lr=2e-3),
                                dict(scheme="W4A16", group_size=32, momentum=0.9, iters=6),
                                dict(scheme="W8A16", group_size=0, iters=4),
                                dict(scheme="W2A16G32", iters=6, dynamic_max_gap=1),
                                dict(scheme="W4A16", group_size=32, layer_config={"model.layers.0.mlp.down_proj": {"bits": 16},
                                                                                  "q_proj": {"bits": 8}, "layers.1.mlp": {"group_size": 64}}),
                                dict(scheme="W4A16", group_size=32, nsamples=6, iters=5),
                                dict(scheme="W4A16", group_size=32, nsamples=3),
                                dict(scheme="W4A16", group_size=32, seed=7),
                                dict(scheme="W4A16", group_size=32, trailing_repeats=True),
                                dict(scheme="W4A16", group_size=32, trailing_repeats=True, pad_token_id=3)],
                         ids=["w4g32", "w2g32_asym", "mxfp4", "w2g32_alg_ext", "nvfp4_alg_ext", "w2g32_asym_alg_ext_baseline_cfg2", "mixtral_w4g32",
                              "mixtral_nvfp4_idle_experts", "opt_w4g32", "opt_w2g32_asym", "gpt2_conv1d_w4g32", "qwen2_w4g32", "qwen3_w4g32", "qwen3_moe_w4g32", "w4a8_int_act", "int8_w8a8", "w3g32", "w8g32", "grad_accumulate_2", "last_iterate",
                              "no_minmax_tuning", "fp_input_chain", "explicit_lrs", "momentum_0.9", "per_tensor_groups", "early_stop", "mixed_layer_config",
                              "ragged_last_batch", "fewer_samples_than_batch", "other_seed", "trailing_repeats_count_as_padding",
                              "pad_token_id_masks_pads"])
def test_block_by_block_pipeline_equals_reference_front_door(kw, tmp_path, monkeypatch):
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round import AutoRound

    from auto_round_amd.schemes import apply_scheme, resolve_scheme

    monkeypatch.chdir(tmp_path)                      # the reference writes ./ar_work_space
    kw = dict(kw)
    moe = kw.pop("moe", False)
    arch = kw.pop("arch", "llama")
    moe = moe or kw.pop("moe_arch", False)
    if arch in ("qwen2", "qwen3", "qwen3_moe"):
        base = _tiny_hf(arch)
    else:
        base = (_tiny_moe(*moe) if isinstance(moe, tuple) else _tiny_moe()) if moe else ({"opt": _tiny_opt, "gpt2": _tiny_gpt2, "llama": _tiny}[arch]())
    nsamples, seed, layer_config = kw.pop("nsamples", 8), kw.pop("seed", 42), kw.pop("layer_config", None)
    tokens = torch.randint(0, 64, (nsamples, 16), generator=torch.Generator().manual_seed(1))
    pad_token_id = kw.pop("pad_token_id", None)
    if kw.pop("trailing_repeats", False):        # samples that end in a run of one token, and a few pad ids in the middle
        tokens[1, -5:] = tokens[1, -1]
        tokens[4, -2:] = tokens[4, -1]
        tokens[6, :] = 3
        tokens[2, 4], tokens[5, 9] = 3, 3
    iters, bs, S = kw.pop("iters", 3), 4, 16
    loop_kw = {k: kw.pop(k) for k in ("gradient_accumulate_steps", "not_use_best_mse", "enable_minmax_tuning", "lr", "minmax_lr",
                                      "dynamic_max_gap", "momentum") if k in kw}
    quanted_input = kw.pop("enable_quanted_input", True)
    if loop_kw.get("gradient_accumulate_steps", 1) != 1:
        bs = 2                                       # 2 micro-batches of 2 = the same global batch of 4

    # --- the reference, front door to tuned weights
    m_ref = copy.deepcopy(base)
    tok = _StubTokenizer()
    tok.pad_token_id = pad_token_id
    ar = AutoRound(m_ref, tokenizer=tok, iters=iters, nsamples=nsamples, seqlen=S, dataset=_Loader(tokens), device_map="cpu",
                   batch_size=bs, enable_torch_compile=False, enable_quanted_input=quanted_input, seed=seed,
                   layer_config=None if layer_config is None else {k: dict(v) for k, v in layer_config.items()}, **loop_kw, **kw)
    q_ref, _ = ar.quantize()

    # --- the same flow with the restatement
    m = copy.deepcopy(base)
    if moe:      # the reference unfuses transformers' 3-D expert parameters itself; here the product's preparation step does
        from auto_round_amd.moe_unfuse import unfuse_moe_experts

        assert len(unfuse_moe_experts(m)) == 2
    for p in m.parameters():
        p.requires_grad_(False)
    blocks = list(_layers(m))
    alg_ext = bool(kw.get("enable_alg_ext", False))
    sch = resolve_scheme(**{k: v for k, v in kw.items() if k != "enable_alg_ext"})
    if (sch.get("act_bits") or 16) <= 8:      # unset activation fields follow the weights', as in the product's front door
        for k, v in (("act_data_type", sch["data_type"]), ("act_sym", sch["sym"]), ("act_dynamic", True), ("act_group_size", sch["group_size"])):
            if sch.get(k) is None:
                sch[k] = v
    from auto_round_amd.autoround import _block_layer_config

    for i, b in enumerate(blocks):
        name = f"{'model.layers' if hasattr(m, 'model') and not hasattr(m.model, 'decoder') else 'blocks'}.{i}"
        apply_scheme(b, sch, layer_config=_block_layer_config(layer_config, name, b))
    _, n_filled = run_flow(m, blocks, tokens, sch, iters=iters, bs=bs, alg_ext=alg_ext, moe=bool(moe), reference_mask=True,
                           quanted_input=quanted_input, tune_kw=loop_kw, seed=seed, pad_token_id=pad_token_id)

    from transformers.pytorch_utils import Conv1D

    lin_ref = {n: p for n, p in _layers(q_ref).named_modules() if isinstance(p, (torch.nn.Linear, Conv1D))}
    lin_mine = {n: p for n, p in _layers(m).named_modules() if isinstance(p, (torch.nn.Linear, Conv1D))}
    assert set(lin_ref) == set(lin_mine) and len(lin_ref) >= 8
    if isinstance(moe, tuple):
        assert n_filled > 0, "the case is meant to contain experts without calibration tokens"
    for n, p1 in lin_ref.items():
        assert torch.equal(p1.weight.view(torch.int16), lin_mine[n].weight.view(torch.int16)), n


@pytest.mark.parametrize("kw", [dict(scheme="W4A16", group_size=32), dict(scheme="W2A16G32", sym=False), dict(scheme="W4A16", group_size=32, sym=False),
                                dict(scheme="W3A16", group_size=32), dict(scheme="MXFP4"), dict(scheme="NVFP4"),
                                dict(scheme="W4A16", group_size=32, format="auto_gptq"), dict(scheme="W4A16", group_size=32, sym=False, format="auto_awq"),
                                dict(scheme="W2A16G32", enable_alg_ext=True)],
                         ids=["w4g32_sym_gptq_words", "w2g32_asym_plain_words", "w4g32_asym_awq_words", "w3g32_sym", "mxfp4_nibbles",
                              "nvfp4_nibbles_and_scales", "format_auto_gptq", "format_auto_awq", "w2g32_alg_ext"])
def test_reference_checkpoint_tensors_equal_the_oracle_packers_on_the_restated_flow(kw, tmp_path, monkeypatch):
    """North-star: "quantized integer weights and packed buffers must match the reference bit-exactly on the same seed/inputs".
    The reference tunes AND saves (format auto_round) on CPU; the restated flow tunes the same model and the C oracle's packers
    -- the ones the HIP packers are held to on the GPU -- pack it: every packed tensor of every block layer is identical."""
    import numpy as np
    from safetensors import safe_open

    from oracle import oracle as orc

    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round import AutoRound

    from auto_round_amd.schemes import apply_scheme, resolve_scheme

    monkeypatch.chdir(tmp_path)
    kw = dict(kw)
    fmt, alg_ext = kw.pop("format", "auto_round"), bool(kw.get("enable_alg_ext", False))
    base = _tiny()
    tokens = torch.randint(0, 64, (8, 16), generator=torch.Generator().manual_seed(1))
    iters, bs, S = 3, 4, 16
    ar = AutoRound(copy.deepcopy(base), tokenizer=_StubTokenizer(), iters=iters, nsamples=8, seqlen=S, dataset=_Loader(tokens),
                   device_map="cpu", batch_size=bs, enable_torch_compile=False, **kw)
    out = str(tmp_path / "ref")
    ar.quantize_and_save(out, format=fmt)
    sub = [d for d in os.listdir(out) if os.path.isdir(os.path.join(out, d))]
    out = os.path.join(out, sub[0]) if sub else out
    ref_t = {}
    for f in os.listdir(out):
        if f.endswith(".safetensors"):
            with safe_open(os.path.join(out, f), "pt") as sf:
                for k in sf.keys():
                    ref_t[k] = sf.get_tensor(k)

    m = copy.deepcopy(base)
    for p in m.parameters():
        p.requires_grad_(False)
    sch = resolve_scheme(**{k: v for k, v in kw.items() if k != "enable_alg_ext"})
    blocks = list(m.model.layers)
    for b in blocks:
        apply_scheme(b, sch)
    run_flow(m, blocks, tokens, sch, iters=iters, bs=bs, reference_mask=True, alg_ext=alg_ext)

    bits_, gs, sym, mx = int(sch["bits"]), int(sch["group_size"]), bool(sch["sym"]), str(sch["data_type"]).startswith("mx")
    nv = str(sch["data_type"]).startswith("nv")
    n_checked = 0
    for name, lin in m.named_modules():
        if not (isinstance(lin, torch.nn.Linear) and hasattr(lin, "scale") and name.startswith("model.layers")):
            continue
        name = name.replace(".orig_layer", "")          # A4 schemes leave the activation-quant shell around the layer
        out_f, in_f = lin.weight.shape
        Wb = orc.to_bits(lin.weight.data).reshape(-1)
        if mx:
            packed, sb = orc.pack_fp4(Wb, orc.to_bits(lin.scale.to(lin.weight.dtype)).reshape(-1), out_f, in_f, gs, 0)
            assert np.array_equal(packed, ref_t[f"{name}.weight_packed"].numpy()), name
            assert np.array_equal(sb, ref_t[f"{name}.weight_scale"].numpy().reshape(sb.shape)), name
        elif nv:      # e4m3 group scales, the (q/k/v- and gate/up-unified) global scale and the static input scale from act_max
            gsc = float(lin.weight_global_scale)
            packed, sb = orc.pack_fp4(Wb, lin.scale.float().numpy().reshape(-1), out_f, in_f, gs, 1, global_scale=gsc)
            assert np.array_equal(packed, ref_t[f"{name}.weight_packed"].numpy()), name
            assert np.array_equal(sb, ref_t[f"{name}.weight_scale"].view(torch.uint8).numpy().reshape(sb.shape)), name
            assert np.float32(gsc) == ref_t[f"{name}.weight_global_scale"].float().numpy().reshape(-1)[0], name
            amax = np.float32(torch.as_tensor(lin.act_max).float().abs().max().item())
            assert np.float32(448.0 * 6.0) * (np.float32(1.0) / amax) == ref_t[f"{name}.input_global_scale"].float().numpy().reshape(-1)[0], name
        else:
            sb = orc.to_bits(lin.scale).reshape(-1)
            zp = float(lin.zp) if not isinstance(lin.zp, torch.Tensor) else lin.zp.float().numpy()
            if sym or fmt == "auto_gptq":             # backend auto_round:auto_gptq / format auto_gptq -> the zp-1 packer
                qw, qz, st = orc.pack_int(Wb, sb, zp, out_f, in_f, gs, bits_, zp_off=1)
                if fmt == "auto_gptq":
                    assert np.array_equal(ref_t[f"{name}.g_idx"].numpy(), np.arange(in_f, dtype=np.int32) // gs), name
            elif bits_ == 4:                          # W4 asym -> AWQ GEMM container
                qw, qz, st = orc.pack_awq(Wb, sb, zp, out_f, in_f, gs)
            else:                                     # other asym -> the plain packer
                qw, qz, st = orc.pack_int(Wb, sb, zp, out_f, in_f, gs, bits_, zp_off=0)
            assert np.array_equal(qw, ref_t[f"{name}.qweight"].numpy()), name
            assert np.array_equal(qz, ref_t[f"{name}.qzeros"].numpy()), name
            assert np.array_equal(st, orc.to_bits(ref_t[f"{name}.scales"])), name
        n_checked += 1
    assert n_checked == 14
