"""The plugin boundary against the LIVE reference on CPU (build container only).  The HIP engine cannot run here and the
reference does not exist on the GPU box, so the glue -- `auto_round_amd.plugin`: registry entry, config mapping, what the
reference's orchestrator actually hands to `quantize_block` (per-sample lists in `input_others`), post-conditions the
orchestrator relies on -- is exercised with the torch restatement standing in for the engine behind the SAME interface the
product quantizer has.  The tuned model must equal the reference's own SignRound run."""
import copy
import inspect
import os
import sys

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "auto_round")), reason="reference tree not present (GPU box)")


def _cpu_block_forward(block, x, others, amp_dtype):
    """auto_round_amd.quantizer.block_forward with the autocast device swapped for the CPU."""
    others = dict(others or {})
    positional = others.pop("positional_inputs", None) or ()
    names = [p for p in inspect.signature(block.forward).parameters.keys() if p != "self"]
    others.setdefault(names[0], x)
    for i, val in enumerate(positional):
        if i + 1 < len(names) and names[i + 1] not in others:
            others[names[i + 1]] = val
    with torch.autocast("cpu", dtype=amp_dtype):
        out = block(**others)
    return out[0] if isinstance(out, (tuple, list)) else out


class _StandInEngine:
    """Same constructor, `quantize_block` signature and `last_stats` as auto_round_amd.quantizer.SignRoundQuantizer."""
    seen = []
    alg_ext = False

    def __init__(self, config, device="cuda"):
        self.config, self.device = config, device

    def quantize_block(self, block, fp_inputs, input_others, fp_outputs, q_inputs=None, block_ctx=None, input_ids=None, **kw):
        from auto_round_amd.quantizer import normalize_input_others, stack_samples
        from oracle import torch_ref as tr

        cfg = self.config
        X = stack_samples(q_inputs if (q_inputs is not None and cfg.enable_quanted_input) else fp_inputs, "cpu")
        Y = stack_samples(fp_outputs, "cpu")
        shared, per_sample = normalize_input_others(input_others, X.shape[0], "cpu")
        _StandInEngine.seen.append(dict(cfg=cfg, others_in={k: type(v).__name__ for k, v in input_others.items()},
                                    shared={k: (tuple(v.shape) if isinstance(v, torch.Tensor) else type(v).__name__) for k, v in shared.items()},
                                    per_sample=list(per_sample), n=X.shape[0], has_q=q_inputs is not None))
        assert not per_sample                      # fixed-seqlen calibration: every sample carries the same mask
        best, info = tr.tune_block(block, X, Y, shared, iters=cfg.iters, batch_size=cfg.batch_size, lr=cfg.lr, minmax_lr=cfg.minmax_lr,
                                   enable_minmax_tuning=cfg.enable_minmax_tuning, input_ids=input_ids, amp_dtype=cfg.amp_dtype,
                                   forward=lambda b, x, o: _cpu_block_forward(b, x, o, cfg.amp_dtype), alg_ext=self.alg_ext)
        from auto_round_amd.wrapper import WrapperWALayer, _set_module

        for n, m in list(block.named_modules()):        # leave what the product's unwrapper leaves: its own activation-quant shell
            if isinstance(m, tr.RefWALayer):
                _set_module(block, n, WrapperWALayer(m.orig_layer))
        self.last_stats = dict(init_loss=info["losses"][0], best_loss=info["best_loss"], best_iter=info["best_iter"],
                               quantized=len(info["names"]), unquantized=0)
        return best


@pytest.mark.parametrize("kw", [dict(scheme="W4A16", group_size=32), dict(scheme="W2A16G32", sym=False),
                                dict(scheme="W2A16G32", enable_alg_ext=True), dict(scheme="MXFP4"), dict(scheme="W4A16", group_size=32, act_bits=8)],
                         ids=["w4g32", "w2g32_asym", "w2g32_alg_ext", "mxfp4_act_quant_shells", "w4a8_act_quant_shells"])
def test_reference_front_door_with_the_plugin_quantizer(kw, tmp_path, monkeypatch):
    shim = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "ref_shim")
    sys.dont_write_bytecode = True
    for p in (shim, REF):
        if p not in sys.path:
            sys.path.insert(0, p)
    from auto_round import AutoRound

    import auto_round_amd.plugin as plugin
    import auto_round_amd.quantizer as product
    from test_pipeline_vs_reference import _Loader, _StubTokenizer, _tiny

    monkeypatch.chdir(tmp_path)
    Cfg, _ = plugin.register()
    _StandInEngine.seen = []
    monkeypatch.setattr(product, "SignRoundQuantizer", _StandInEngine)

    class _V2StandIn(product.SignRoundV2Quantizer):          # keeps the product's imatrix hooks and scheme rules; engine swapped
        alg_ext = True

        def __init__(self, config, device="cuda"):
            self.config, self.device, self._scheme = config, device, None

        quantize_block = _StandInEngine.quantize_block

    monkeypatch.setattr(product, "SignRoundV2Quantizer", _V2StandIn)
    base = _tiny()
    tokens = torch.randint(0, 64, (8, 16), generator=torch.Generator().manual_seed(1))
    common = dict(tokenizer=_StubTokenizer(), nsamples=8, seqlen=16, dataset=_Loader(tokens), device_map="cpu", batch_size=4,
                  enable_torch_compile=False, **kw)
    q_ref, _ = AutoRound(copy.deepcopy(base), iters=3, **common).quantize()                       # the reference's own SignRound
    q_plug, _ = AutoRound(copy.deepcopy(base), alg_configs=Cfg(iters=3), **common).quantize()     # same front door, plugin quantizer

    assert len(_StandInEngine.seen) == 2                                   # one call per block, through plugin.quantize_block
    first, second = _StandInEngine.seen
    assert first["cfg"].iters == 3 and first["cfg"].batch_size == 4 and abs(first["cfg"].lr - 1.0 / 3) < 1e-12
    assert first["cfg"].amp_dtype == torch.bfloat16 and first["cfg"].enable_quanted_input is True
    assert first["others_in"]["attention_mask"] == "list" and first["shared"]["attention_mask"] == (1, 1, 16, 16)
    assert first["n"] == 8 and not first["has_q"] and second["has_q"]      # block 1 is tuned on block 0's quantised output
    from auto_round.wrapper import WrapperWALayer as RefShell

    shells_ref = sorted(n for n, m in q_ref.model.layers.named_modules() if isinstance(m, RefShell))
    shells_plug = sorted(n for n, m in q_plug.model.layers.named_modules() if isinstance(m, RefShell))
    assert shells_ref == shells_plug and (len(shells_ref) == 14) == ((kw.get("act_bits") or 16) <= 8 or kw["scheme"] == "MXFP4")
    lin_ref = {n: p for n, p in q_ref.model.layers.named_modules() if isinstance(p, torch.nn.Linear)}
    lin_plug = {n: p for n, p in q_plug.model.layers.named_modules() if isinstance(p, torch.nn.Linear)}
    assert set(lin_ref) == set(lin_plug) and len(lin_ref) == 14
    for n, p in lin_ref.items():
        assert torch.equal(p.weight.view(torch.int16), lin_plug[n].weight.view(torch.int16)), n
        assert torch.equal(p.scale.float(), lin_plug[n].scale.float()), n  # post-conditions the exporters consume
        zr, zq = getattr(p, "zp", None), getattr(lin_plug[n], "zp", None)
        assert (zr == zq) if not isinstance(zr, torch.Tensor) else torch.equal(zr, zq), n


def test_normalize_input_others():
    from auto_round_amd.quantizer import normalize_input_others

    N = 4
    mask = [torch.ones(1, 1, 5, 5) for _ in range(N)]
    pe = [(torch.zeros(1, 5, 8), torch.ones(1, 5, 8)) for _ in range(N)]
    others = {"positional_inputs": [], "attention_mask": mask, "position_embeddings": pe, "position_ids": [torch.arange(5).view(1, 5)] * N,
              "past_key_values": None, "use_cache": False}
    shared, per = normalize_input_others(others, N, "cpu")
    assert per == {} and tuple(shared["attention_mask"].shape) == (1, 1, 5, 5) and isinstance(shared["position_embeddings"], tuple)
    assert tuple(shared["position_ids"].shape) == (1, 5) and shared["past_key_values"] is None and shared["use_cache"] is False
    mask[2] = mask[2] * 0                                                   # samples that differ keep their own rows
    shared, per = normalize_input_others(others, N, "cpu")
    assert "attention_mask" not in shared and tuple(per["attention_mask"].shape) == (4, 1, 5, 5)
    assert torch.equal(per["attention_mask"].index_select(0, torch.tensor([2, 0]))[0], mask[2][0])
    standalone = {"attention_mask": torch.ones(1, 1, 5, 5), "position_embeddings": (torch.zeros(1, 5, 8),)}
    same, per = normalize_input_others(standalone, N, "cpu")
    assert same is standalone and per == {}                                 # the standalone front door's dict is left alone
    with pytest.raises(ValueError):
        normalize_input_others({"attention_mask": [torch.ones(2, 1, 5, 5)] * N}, N, "cpu")
