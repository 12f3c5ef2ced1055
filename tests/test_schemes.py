"""Host logic of the standalone front door: scheme presets equal the reference's, scheme application, block discovery."""
import os
import sys

import pytest
import torch

from auto_round_amd import schemes as S

REF = "/root/reference"


def test_presets_resolve_and_override():
    w4 = S.resolve_scheme("w4a16")
    assert (w4["bits"], w4["group_size"], w4["sym"], w4["data_type"], w4["act_bits"]) == (4, 128, True, "int", 16)
    w2 = S.resolve_scheme("W2A16G32", sym=False)                # BASELINE cfg 3: W2 g32 asym
    assert (w2["bits"], w2["group_size"], w2["sym"]) == (2, 32, False)
    nv = S.resolve_scheme("NVFP4")
    assert nv["act_data_type"] == "nv_fp4_with_static_gs" and nv["group_size"] == 16 and nv["act_bits"] == 4
    assert S.resolve_scheme({"bits": 3, "group_size": 64, "sym": True, "data_type": "int"}, bits=4)["bits"] == 4
    with pytest.raises(ValueError):
        S.resolve_scheme("FP8_STATIC")
    with pytest.raises(KeyError):
        S.resolve_scheme("W4A16", super_bits=6)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
def test_presets_equal_the_reference_presets():
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.schemes import PRESET_SCHEMES

    for name, mine in S.PRESET_SCHEMES.items():
        ref = PRESET_SCHEMES[name]
        for k in ("bits", "group_size", "sym", "data_type", "act_bits"):
            assert getattr(ref, k) == mine[k], (name, k)
        if mine["act_bits"] < 16:
            for k in ("act_data_type", "act_group_size", "act_sym", "act_dynamic"):
                assert getattr(ref, k) == mine[k], (name, k)


def test_apply_scheme_and_block_discovery():
    from auto_round_amd.autoround import get_block_names

    class Blk(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.q_proj = torch.nn.Linear(64, 64, bias=False)
            self.odd = torch.nn.Linear(64, 48, bias=False)       # 48 % 32 != 0 -> stays 16 bit
            self.mlp = torch.nn.Module()
            self.mlp.gate = torch.nn.Linear(64, 8, bias=False)   # MoE router -> stays 16 bit
            self.mlp.up = torch.nn.Linear(64, 128, bias=False)

    class M(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.embed = torch.nn.Embedding(10, 64)
            self.model = torch.nn.Module()
            self.model.layers = torch.nn.ModuleList([Blk(), Blk(), Blk()])
            self.vision = torch.nn.ModuleList([Blk()])
            self.lm_head = torch.nn.Linear(64, 10, bias=False)

    m = M()
    groups = get_block_names(m)
    assert ["model.layers.0", "model.layers.1", "model.layers.2"] in groups and ["vision.0"] in groups
    cfg = S.apply_scheme(m.model.layers[0], S.resolve_scheme("W4A16", group_size=32), layer_config={"mlp.up": {"bits": 8}})
    b = m.model.layers[0]
    assert b.q_proj.bits == 4 and b.q_proj.group_size == 32 and b.q_proj.sym and b.q_proj.scale_dtype == torch.float16
    assert b.odd.bits == 16 and b.mlp.gate.bits == 16 and b.mlp.up.bits == 8
    assert cfg["mlp.up"]["bits"] == 8 and cfg["odd"]["bits"] == 16 and set(cfg) == {"q_proj", "odd", "mlp.gate", "mlp.up"}


def test_front_door_fails_loudly_without_a_hip_device():
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    from auto_round_amd.autoround import AutoRound

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AutoRound(torch.nn.Linear(32, 32), None, dataset=torch.zeros(1, 8, dtype=torch.long))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        AutoRound(torch.nn.Linear(32, 32), None, device_map="cpu", dataset=torch.zeros(1, 8, dtype=torch.long))
    # the reference's memory / compile knobs, its legacy `device` alias and an explicit SignRound request are accepted
    # (the call gets as far as the device check); anything that would change the computation is refused up front
    for extra in (dict(low_gpu_mem_usage=True, enable_torch_compile=False, low_cpu_mem_usage=True), dict(device="cuda:0"),
                  dict(device_map="auto"), dict(device_map="0"), dict(algorithm="sign_round"), dict(platform="hf")):
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            AutoRound(torch.nn.Linear(32, 32), None, dataset=torch.zeros(1, 8, dtype=torch.long), **extra)
    with pytest.raises(NotImplementedError, match="SignRound"):
        AutoRound(torch.nn.Linear(32, 32), None, algorithm="awq")
    with pytest.raises(NotImplementedError, match="platform"):
        AutoRound(torch.nn.Linear(32, 32), None, platform="model_scope")
    with pytest.raises(TypeError, match="outside the MI355X hot path"):
        AutoRound(torch.nn.Linear(32, 32), None, quant_lm_head=True)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
def test_learning_rate_rules_equal_the_reference_config():
    """auto lr = 1/iters, 2/iters for <= 3 bits at >= 1000 iterations; explicit lr wins; minmax_lr falls back to lr
    (sign_round/config.py:110-140)."""
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round.algorithms.quantization.sign_round.config import SignRoundConfig as RefCfg

    from auto_round_amd.quantizer import SignRoundConfig

    for iters in (1, 50, 200, 999, 1000, 2000):
        for kw in ({}, {"lr": 0.003}, {"minmax_lr": 0.002}, {"lr": 0.004, "minmax_lr": 0.001}):
            ref, mine = RefCfg(iters=iters, **kw), SignRoundConfig(iters=iters, **kw)
            for bits in (2, 3, 4, 8, None):
                assert ref.compute_lr(bits) == mine.compute_lr(bits), (iters, kw, bits)
                assert ref.compute_minmax_lr(bits) == mine.compute_minmax_lr(bits), (iters, kw, bits)


def _ref_paths():
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")
def test_host_helpers_equal_the_reference_helpers():
    """check_need_act_calibration (truth table), get_block_names (Llama, Mixtral) and the NVFP4 block-wise global-scale
    unification (q/k/v and gate/up share the minimum) against the reference's own functions."""
    _ref_paths()
    from auto_round.compressors.utils import check_need_act_calibration as ref_need
    from auto_round.data_type.utils import update_block_global_scale_if_needed as ref_update
    from auto_round.utils import get_block_names as ref_blocks
    from transformers import LlamaConfig, LlamaForCausalLM

    from auto_round_amd.autoround import get_block_names
    from auto_round_amd.quantizer import check_need_act_calibration
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from auto_round_amd.wrapper import update_block_global_scale_if_needed

    for dyn in (True, False, None):
        for adt in (None, "int", "mx_fp", "nv_fp4_with_static_gs", "fp8_static"):
            for bits in (4, 8, 16, None):
                assert check_need_act_calibration(dyn, adt, bits) == ref_need(dyn, adt, bits), (dyn, adt, bits)

    torch.manual_seed(0)
    m = LlamaForCausalLM(LlamaConfig(hidden_size=64, intermediate_size=128, num_attention_heads=4, num_key_value_heads=2,
                                     num_hidden_layers=3, vocab_size=64)).to(torch.bfloat16)
    assert get_block_names(m) == ref_blocks(m)

    import copy

    blk_a, blk_b = copy.deepcopy(m.model.layers[0]), copy.deepcopy(m.model.layers[0])
    sch = resolve_scheme("NVFP4")
    apply_scheme(blk_a, sch)
    apply_scheme(blk_b, sch)
    update_block_global_scale_if_needed(blk_a)
    ref_update(blk_b, "nv_fp", 16)
    for (n, a), (_, b) in zip(blk_a.named_modules(), blk_b.named_modules()):
        if isinstance(a, torch.nn.Linear):
            assert torch.equal(a.weight_global_scale.reshape(-1).float(), b.weight_global_scale.reshape(-1).float()), n
    qa, ka, va = blk_a.self_attn.q_proj, blk_a.self_attn.k_proj, blk_a.self_attn.v_proj
    assert float(qa.weight_global_scale) == float(ka.weight_global_scale) == float(va.weight_global_scale)


def test_loss_mask_ids_known_answers():
    """Which positions enter the loss (reference: calibration/llm.py:341-360): pads by id, or trailing repeats of the last token when
    the tokenizer has no pad id; always without the last position."""
    from auto_round_amd.autoround import loss_mask_ids

    t = torch.tensor([[5, 6, 7, 7, 7], [1, 2, 3, 4, 5], [9, 9, 9, 9, 9], [0, 3, 0, 2, 0]])
    assert loss_mask_ids(t).tolist() == [[5, 6, -100, -100, -100], [1, 2, 3, 4, -100], [-100] * 5, [0, 3, 0, 2, -100]]
    assert loss_mask_ids(t, pad_token_id=0).tolist() == [[5, 6, 7, 7, -100], [1, 2, 3, 4, -100], [9, 9, 9, 9, -100],
                                                         [-100, 3, -100, 2, -100]]
    assert t[0, 2] == 7                                  # the input is left alone
