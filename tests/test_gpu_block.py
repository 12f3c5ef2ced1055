"""Block-level parity on the GPU: auto_round_amd's quantize_block (HIP path) against oracle/torch_ref.tune_block
(the pinned torch restatement of the reference loop) run on the SAME device with the SAME GEMM library, seeds and
index schedule.  Integer/baked results are compared exactly at iteration granularity where the arithmetic is
order-free, and statistically where the reference itself is not bit-reproducible (trajectories
diverge with GEMM summation order because sign-SGD is chaotic)."""
import copy
import random

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def make_layer(kind="llama", bits=4, gs=32, sym=True, seed=0, hidden=256, ffn=512):
    import __graft_entry__ as ge

    if kind == "llama":
        layer, rope, cfg = ge._tiny_llama_layer(hidden=hidden, ffn=ffn, seed=seed, bits=bits, gs=gs, sym=sym)
        return layer.cuda(), rope.cuda(), cfg
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer

    torch.manual_seed(seed)
    cfg = OPTConfig(hidden_size=hidden, ffn_dim=ffn, num_attention_heads=4, num_hidden_layers=1, vocab_size=128,
                    max_position_embeddings=128, word_embed_proj_dim=hidden)
    cfg._attn_implementation = "sdpa"
    layer = OPTDecoderLayer(cfg).to(torch.bfloat16).eval()
    for p in layer.parameters():
        p.requires_grad_(False)
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            m.bits, m.group_size, m.sym, m.data_type, m.scale_dtype, m.act_bits = bits, gs, sym, "int", torch.float16, 16
    return layer.cuda(), None, cfg


def make_data(rope, cfg, N=16, S=32, seed=1):
    g = torch.Generator().manual_seed(seed)
    X = torch.randn(N, S, cfg.hidden_size, generator=g).to(torch.bfloat16).cuda()
    others = {}
    if rope is not None:
        pos = torch.arange(S, device="cuda").unsqueeze(0)
        cos, sin = rope(X[:1], pos)
        others = {"position_embeddings": (cos, sin), "attention_mask": None, "position_ids": pos}
    return X, others


def fwd(blk, x, others):
    out = blk(x, **others)
    return out[0] if isinstance(out, (tuple, list)) else out


def targets(layer, X, others, bs=4):
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        return torch.cat([fwd(layer, X[i:i + bs], others) for i in range(0, X.shape[0], bs)])


def linears(block):
    return {n: m for n, m in block.named_modules() if isinstance(m, torch.nn.Linear)}


@pytest.mark.parametrize("kind,bits,gs,sym", [("llama", 4, 32, True), ("opt", 4, 128, True), ("llama", 2, 32, False),
                                              ("llama", 8, 0, True), ("opt", 8, 0, False)])      # gs 0: per-tensor groups
def test_quantize_block_vs_torch_ref_loop(kind, bits, gs, sym):
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer(kind, bits, gs, sym)
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 4, 4

    blk_o = copy.deepcopy(layer)
    random.seed(11)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd)

    blk_m = copy.deepcopy(layer)
    random.seed(11)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=bits), device="cuda")
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    # iteration 0 sees identical parameters (V=0, scales=1): identical fake-quant weights, same GEMMs -> same loss
    assert abs(st["init_loss"] - info["losses"][0]) <= 2e-3 * info["losses"][0], (st, info["losses"])
    assert abs(st["best_loss"] - info["best_loss"]) <= 2e-2 * info["best_loss"]
    lo, lm = linears(blk_o), linears(blk_m)
    agree = []
    for n in lo:
        assert tuple(lm[n].scale.shape) == tuple(lo[n].scale.shape) and lm[n].scale.dtype == torch.float16
        assert lm[n].scale.device.type == "cpu"
        if sym:
            assert lm[n].zp == lo[n].zp == 2 ** (bits - 1)
        else:
            assert tuple(lm[n].zp.shape) == tuple(lo[n].zp.shape)
        agree.append((lm[n].weight == lo[n].weight).float().mean().item())
        # V moved by at most sum(lr) and only in multiples consistent with sign steps
        assert float(best_m[n]["value"].abs().max()) <= sum(tr.linear_lr_stream(1.0 / iters, iters)) + 1e-6
        if gs == 0:     # the reference's per-tensor shapes: value [1, numel], one min / max scale, scale stored flat
            assert tuple(best_m[n]["value"].shape) == (1, lm[n].weight.numel()) and tuple(best_m[n]["min_scale"].shape) == (1,)
            assert tuple(lm[n].scale.shape) == (1,)
    assert np.mean(agree) > 0.97, agree


def test_first_step_is_bit_exact_vs_torch_ref():
    """One iteration, no best-tracking subtleties: after step 0 every V/min/max equals the autograd + SignSGD result
    computed by torch on the same device (dWq from the same GEMM; group sums only matter in sign)."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=3)
    X, others = make_data(rope, cfg, N=8)
    Y = targets(layer, X, others)
    rec = {}

    def record(i, wrappers, total):
        rec["grads"] = {n: {k: p.grad.clone() for k, p in w.params.items()} for n, w in wrappers.items()}

    blk_o = copy.deepcopy(layer)
    random.seed(5)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=200, batch_size=4, forward=fwd, record=record, max_iters_to_run=1)
    # params after one step with lr0 = 1/200 : -lr0 * sign(grad)
    blk_m = copy.deepcopy(layer)
    random.seed(5)
    q = SignRoundQuantizer(SignRoundConfig(iters=200, batch_size=4, bits=4, not_use_best_mse=True), device="cuda")
    # exactly ONE applied step: with not_use_best_mse the parameters are snapshotted at the last iteration BEFORE its
    # optimizer step (reference sign_round/quantizer.py:513-514), so two iterations leave the state after step 0
    q.config.iters = 2
    q.config.lr = 1.0 / 200
    q.config.minmax_lr = 1.0 / 200
    q.config.lr_is_auto = False
    q.config.minmax_lr_is_auto = False
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    lr0 = np.float32(1.0 / 200)
    tot = mism = 0
    for n, gd in rec["grads"].items():
        v_exp = -lr0 * torch.sign(gd["value"])
        v_got = best_m[n]["value"]
        tot += v_exp.numel()
        mism += int((v_exp != v_got).sum())
        for k in ("min_scale", "max_scale"):
            exp = 1.0 - lr0 * torch.sign(gd[k])
            assert (exp == best_m[n][k]).float().mean() > 0.98, (n, k)
    # dWq comes from torch.mm(dY^T, X) here and from autograd's linear backward there: same hipBLASLt GEMM in
    # practice; allow a vanishing fraction of sign flips of near-zero gradients
    assert mism / tot < 2e-3, (mism, tot)


def test_post_conditions_and_options():
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.wrapper import WrapperLinear

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=2)
    X, others = make_data(rope, cfg, N=8)
    W0 = {n: m.weight.detach().clone() for n, m in linears(layer).items()}
    for kw in (dict(), dict(enable_minmax_tuning=False), dict(not_use_best_mse=True), dict(gradient_accumulate_steps=2),
               dict(dynamic_max_gap=1), dict(fuse_next_forward=False)):
        blk = copy.deepcopy(layer)
        random.seed(0)
        q = SignRoundQuantizer(SignRoundConfig(iters=5, batch_size=2, bits=4, **kw), device="cuda")
        fp_out, q_out, best = q.compress_block(blk, X, others)
        assert not any(isinstance(m, WrapperLinear) for m in blk.modules()), kw
        assert not hasattr(blk, "_ar_arenas")
        st = q.last_stats
        assert st["quantized"] == 7 and np.isfinite(st["best_loss"]), (kw, st)
        if kw.get("dynamic_max_gap"):
            assert st["iters_run"] <= 5
        for n, m in linears(blk).items():
            s = m.scale.float().cuda().repeat_interleave(32, 1)
            qint = torch.round(m.weight.float() / s)
            assert float(qint.min()) >= -8 and float(qint.max()) <= 7, (kw, n)
            assert torch.equal((s * qint).to(torch.bfloat16), m.weight), (kw, n)
            if kw.get("enable_minmax_tuning") is False:
                assert "min_scale" not in best[n]
            assert not torch.equal(m.weight, W0[n])
        # never worse than plain RTN (V=0, scales=1) on the calibration data: iteration 0 IS RTN and the best iterate wins
        if not kw:
            blk_rtn = copy.deepcopy(layer)
            q0 = SignRoundQuantizer(SignRoundConfig(iters=0, batch_size=2, bits=4), device="cuda")
            _, rtn_out, _ = q0.compress_block(blk_rtn, X, others)
            e_tuned = (q_out.float() - fp_out.float()).pow(2).mean().item()
            e_rtn = (rtn_out.float() - fp_out.float()).pow(2).mean().item()
            assert e_tuned <= e_rtn, (e_tuned, e_rtn)


def test_tuning_improves_over_rtn():
    """60 iterations at the reference's auto lr (1/iters): the best iterate must beat iteration 0 (== RTN)."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=4)
    X, others = make_data(rope, cfg, N=16)
    random.seed(1)
    q = SignRoundQuantizer(SignRoundConfig(iters=60, batch_size=4, bits=4), device="cuda")
    fp_out, q_out, best = q.compress_block(layer, X, others)
    st = q.last_stats
    assert st["best_iter"] > 0 and st["best_loss"] < st["init_loss"], st
    assert st["n_improved"] >= 2


def test_smoke_entry():
    import __graft_entry__ as ge

    ge.smoke()


def _set_scheme(layer, data_type, gs, act):
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            m.bits, m.group_size, m.sym, m.data_type = 4, gs, True, data_type
            if act:
                m.act_bits, m.act_group_size, m.act_sym, m.act_dynamic = 4, gs, True, True
                m.act_data_type = "mx_fp" if data_type.startswith("mx") else "nv_fp4_with_static_gs"


@pytest.mark.parametrize("data_type,gs,act", [("mx_fp", 32, False), ("mx_fp", 32, True), ("nv_fp", 16, False), ("nv_fp", 16, True)])
def test_fp4_schemes_vs_torch_ref_loop(data_type, gs, act):
    """MXFP4 / NVFP4 weight tuning (with and without 4-bit activation fake-quant, the cfg-5 schemes) against the torch
    restatement of the reference loop on the same device."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.wrapper import WrapperWALayer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("llama", 4, gs, True, seed=7)
    _set_scheme(layer, data_type, gs, act)
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 4, 4
    blk_o = copy.deepcopy(layer)
    random.seed(3)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd)
    blk_m = copy.deepcopy(layer)
    random.seed(3)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=4), device="cuda")
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert abs(st["init_loss"] - info["losses"][0]) <= 5e-3 * info["losses"][0], (st, info["losses"])
    assert abs(st["best_loss"] - info["best_loss"]) <= 5e-2 * info["best_loss"]
    agree = []
    for (n1, m1), (n2, m2) in zip(blk_o.named_modules(), blk_m.named_modules()):
        if isinstance(m1, torch.nn.Linear):
            assert isinstance(m2, torch.nn.Linear) and m2.zp is None
            assert tuple(m2.scale.shape) == tuple(m1.scale.shape), (n1, m2.scale.shape, m1.scale.shape)
            if data_type == "nv_fp":
                assert float(m2.weight_global_scale) == float(getattr(m1, "weight_global_scale", m2.weight_global_scale))
            agree.append((m1.weight == m2.weight).float().mean().item())
    assert np.mean(agree) > 0.95, agree
    assert any(isinstance(m, WrapperWALayer) for m in blk_m.modules()) == act
    # the unwrapped block (with its activation fake-quant layers) still runs
    with torch.no_grad():
        out = q.forward_all(blk_m, X, others)
    assert torch.isfinite(out.float()).all()


@pytest.mark.parametrize("gs,bits,sym", [(96, 4, True), (-1, 4, False), (80, 2, False)])
def test_padded_and_per_channel_groups_vs_torch_ref(gs, bits, sym):
    """group sizes that do not divide in_features (zero-padded rows) and per-output-channel groups."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("opt", bits, gs, sym, seed=9)     # hidden 256, ffn 512: 96 / 80 divide neither
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 3, 4
    blk_o = copy.deepcopy(layer)
    random.seed(2)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd)
    blk_m = copy.deepcopy(layer)
    random.seed(2)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=bits), device="cuda")
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert abs(st["init_loss"] - info["losses"][0]) <= 2e-3 * info["losses"][0], (st, info["losses"])
    lo, lm = linears(blk_o), linears(blk_m)
    agree = []
    for n in lo:
        assert tuple(lm[n].scale.shape) == tuple(lo[n].scale.shape), (n, lm[n].scale.shape, lo[n].scale.shape)
        assert tuple(best_m[n]["value"].shape) == tuple(best_o[n]["value"].shape)
        agree.append((lm[n].weight == lo[n].weight).float().mean().item())
    assert np.mean(agree) > 0.97, agree


def test_valid_token_mask_matches_torch_ref():
    """input_ids with -100 at the last position of every sample (what the reference's calibrator always produces) and a
    few padded positions: the masked loss / gradient path vs the torch restatement."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=12)
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    ids = torch.randint(0, 100, (X.shape[0], X.shape[1]))
    ids[:, -1] = -100
    ids[3, -5:] = -100
    id_list = list(torch.split(ids, 1, dim=0))
    iters, bs = 3, 4
    blk_o = copy.deepcopy(layer)
    random.seed(4)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd, input_ids=id_list)
    blk_m = copy.deepcopy(layer)
    random.seed(4)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=4), device="cuda")
    best_m = q.quantize_block(blk_m, X, others, Y, None, None, input_ids=id_list)
    st = q.last_stats
    assert abs(st["init_loss"] - info["losses"][0]) <= 2e-3 * info["losses"][0], (st, info["losses"])
    # and the mask really changes the loss normalisation compared to the unmasked path
    blk_u = copy.deepcopy(layer)
    random.seed(4)
    q.quantize_block(blk_u, X, others, Y, None, None)
    assert q.last_stats["init_loss"] > 10 * st["init_loss"]
    agree = [(a.weight == b.weight).float().mean().item() for a, b in zip(linears(blk_o).values(), linears(blk_m).values())]
    assert np.mean(agree) > 0.97, agree


def test_three_block_stack_with_quantised_input_chaining():
    """The per-block driver loop (orchestrator._quantize_blocks): block k+1 is tuned on the QUANTISED output of block k
    against the fp chain; compared with the same loop built from the torch restatement."""
    from auto_round_amd.model_tuner import tune_blocks
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layers = []
    rope = cfg = None
    for k in range(3):
        layer, rope, cfg = make_layer("llama", 4, 32, True, seed=20 + k)
        layers.append(layer)
    X, others = make_data(rope, cfg, N=8)
    iters, bs = 3, 4
    # oracle loop
    ref_blocks = [copy.deepcopy(l) for l in layers]
    random.seed(9)
    fp_in, q_in, ref_losses = X, None, []
    for blk in ref_blocks:
        fp_out = targets(blk, fp_in, others, bs)
        _, info = tr.tune_block(blk, q_in if q_in is not None else fp_in, fp_out, others, iters=iters, batch_size=bs, forward=fwd)
        ref_losses.append(info["losses"][0])
        q_in = targets(blk, q_in if q_in is not None else fp_in, others, bs)
        fp_in = fp_out
    # product loop
    my_blocks = [copy.deepcopy(l) for l in layers]
    random.seed(9)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=4), device="cuda")
    recs = tune_blocks(my_blocks, X, others, q, pack=True)
    assert len(recs) == 3 and all("packed" in r and len(r["packed"]) == 7 for r in recs)
    for r, l0 in zip(recs, ref_losses):
        assert abs(r["stats"]["init_loss"] - l0) <= 2e-2 * l0, (r["stats"], l0)   # later blocks see slightly different q inputs
    agree = []
    for a, b in zip(ref_blocks, my_blocks):
        for (n1, m1), (n2, m2) in zip(linears(a).items(), linears(b).items()):
            agree.append((m1.weight == m2.weight).float().mean().item())
    assert np.mean(agree) > 0.95, agree
    # packed buffers of the first block have the GPTQ shapes
    pk = recs[0]["packed"]["self_attn.q_proj"]
    assert pk.qweight.shape == (256 // 32 * 4, 256) and pk.qweight.dtype == torch.int32
    assert pk.scales.shape == (256 // 32, 256) and pk.qzeros.shape == (256 // 32, 256 // 32 * 4)
    assert bool((pk.qzeros.view(torch.int32) == 0x77777777).all())


@pytest.mark.parametrize("scheme", ["W4A16", "MXFP4", "NVFP4"])
def test_sparse_moe_block_vs_torch_ref(scheme):
    """Mixtral-style block with unfused experts (cfg 5 layout): 4 attention + 3 x E expert linears in one arena,
    token-dependent expert activity (an expert without tokens must not be stepped), optional 4-bit activations."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.testing.moe import build_moe_decoder_layer, set_scheme
    from oracle import torch_ref as tr

    layer, rope, cfg = build_moe_decoder_layer(hidden=256, ffn=512, heads=4, kv_heads=2, num_experts=4, top_k=2, seed=5)
    set_scheme(layer, scheme)
    for m in layer.modules():      # tiny layer: use group sizes that divide 256/512 for the int preset too
        if isinstance(m, torch.nn.Linear) and getattr(m, "bits", 16) < 16 and scheme == "W4A16":
            m.group_size = 32
    X, others = make_data(rope.cuda(), cfg, N=8, S=16)
    Y = targets(layer, X, others, 4)
    iters, bs = 3, 4
    blk_o = copy.deepcopy(layer)
    random.seed(6)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd)
    blk_m = copy.deepcopy(layer)
    random.seed(6)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=4), device="cuda")
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert st["quantized"] == 4 + 3 * 4 and st["unquantized"] == 1, st       # router gate stays fp
    assert abs(st["init_loss"] - info["losses"][0]) <= 5e-3 * info["losses"][0], (st, info["losses"])
    agree = []
    for (n1, m1), (n2, m2) in zip(blk_o.named_modules(), blk_m.named_modules()):
        if isinstance(m1, torch.nn.Linear) and hasattr(m1, "scale"):
            agree.append((m1.weight == m2.weight).float().mean().item())
    assert len(agree) == 16 and np.mean(agree) > 0.95, agree


def test_layer_without_gradient_is_not_stepped():
    """A wrapped linear whose forward is skipped in an iteration (an idle expert) keeps V/min/max unchanged, like a
    parameter with grad None in the reference's SignSGD."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    class TwoPath(torch.nn.Module):
        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.a = torch.nn.Linear(64, 64, bias=False)
            self.b = torch.nn.Linear(64, 64, bias=False)   # never used in forward

        def forward(self, hidden_states):
            return self.a(hidden_states)

    blk = TwoPath().to(torch.bfloat16).cuda()
    for m in (blk.a, blk.b):
        m.bits, m.group_size, m.sym, m.data_type, m.scale_dtype, m.act_bits = 4, 32, True, "int", torch.float16, 16
        m.weight.requires_grad_(False)
    X = torch.randn(8, 4, 64, device="cuda").to(torch.bfloat16)
    Y = (X.float() @ blk.a.weight.float().t()).to(torch.bfloat16)
    random.seed(0)
    q = SignRoundQuantizer(SignRoundConfig(iters=4, batch_size=4, bits=4, not_use_best_mse=True), device="cuda")
    best = q.quantize_block(blk, X, {}, Y, None, None)
    assert float(best["b"]["value"].abs().max()) == 0.0 and bool((best["b"]["min_scale"] == 1).all())
    assert float(best["a"]["value"].abs().max()) > 0.0


def test_edge_cases_empty_ragged_and_unquantized():
    """Empty launches, nsamples not divisible by the batch, batch larger than nsamples, a block without any quantised
    layer, and a GPT-2 style block whose projections are transformers Conv1D modules (weight stored [in, out])."""
    from auto_round_amd import ops
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    # empty group arrays are a no-op, not an error
    e = torch.empty(0, dtype=torch.bfloat16, device="cuda")
    f = torch.empty(0, dtype=torch.float32, device="cuda")
    assert ops.qdq_int_fwd(e, f, e, e, f, f, gs=128, bits=4, sym=True).numel() == 0
    lr = torch.tensor([0.01], device="cuda")
    ops.qdq_int_bwd_sgd_(e, e, f, e, e, f, f, gs=128, bits=4, sym=True, lr_v=lr, lr_mm=lr)

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=31)
    # 10 samples, batch 4: the sampler reshuffles when fewer than a batch remain (compressors/utils.py:388-438)
    X, others = make_data(rope, cfg, N=10)
    blk = copy.deepcopy(layer)
    random.seed(1)
    q = SignRoundQuantizer(SignRoundConfig(iters=7, batch_size=4, bits=4), device="cuda")
    fp_out, q_out, best = q.compress_block(blk, X, others)
    assert fp_out.shape[0] == 10 and q_out.shape[0] == 10 and np.isfinite(q.last_stats["best_loss"])
    # batch larger than nsamples -> clamped to nsamples (quantizer.py:441-442)
    blk = copy.deepcopy(layer)
    q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=64, bits=4), device="cuda")
    q.compress_block(blk, X[:3], others)
    assert np.isfinite(q.last_stats["best_loss"])
    # nothing to quantise: returns {} and leaves the block untouched
    blk = copy.deepcopy(layer)
    for m in blk.modules():
        if isinstance(m, torch.nn.Linear):
            m.bits = 16
    W0 = [m.weight.clone() for m in blk.modules() if isinstance(m, torch.nn.Linear)]
    q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=4, bits=4), device="cuda")
    assert q.quantize_block(blk, X, others, targets(blk, X, others, 5), None, None) == {}
    assert all(torch.equal(a, m.weight) for a, m in zip(W0, [m for m in blk.modules() if isinstance(m, torch.nn.Linear)]))


def test_conv1d_block_gpt2_style():
    from transformers import GPT2Config
    from transformers.models.gpt2.modeling_gpt2 import GPT2Block
    from transformers.pytorch_utils import Conv1D

    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    torch.manual_seed(0)
    cfg = GPT2Config(n_embd=128, n_head=4, n_layer=1, n_positions=64, vocab_size=100)
    cfg._attn_implementation = "sdpa"
    blk = GPT2Block(cfg, layer_idx=0).to(torch.bfloat16).cuda().eval()
    for p in blk.parameters():
        p.requires_grad_(False)
    convs = {n: m for n, m in blk.named_modules() if isinstance(m, Conv1D)}
    assert len(convs) == 4
    for m in convs.values():
        m.bits, m.group_size, m.sym, m.data_type, m.scale_dtype, m.act_bits = 4, 32, True, "int", torch.float16, 16
    X = torch.randn(8, 16, 128, generator=torch.Generator().manual_seed(2)).to(torch.bfloat16).cuda()
    W0 = {n: m.weight.detach().clone() for n, m in convs.items()}
    random.seed(3)
    q = SignRoundQuantizer(SignRoundConfig(iters=20, batch_size=4, bits=4), device="cuda")
    fp_out, q_out, best = q.compress_block(blk, X, {})
    st = q.last_stats
    assert st["quantized"] == 4 and st["best_loss"] <= st["init_loss"]
    for n, m in blk.named_modules():
        if isinstance(m, Conv1D):
            in_f, out_f = m.weight.shape                       # Conv1D stores [in, out]
            assert tuple(m.scale.shape) == (out_f, in_f // 32)
            w2d = m.weight.t().float()                         # [out, in] view the quantizer works on
            s = m.scale.float().cuda().repeat_interleave(32, 1)
            qi = torch.round(w2d / s)
            assert float(qi.min()) >= -8 and float(qi.max()) <= 7
            assert torch.equal((s * qi).to(torch.bfloat16), m.weight.t())
            assert not torch.equal(m.weight, W0[n])


@pytest.mark.parametrize("wdtype,amp", [(torch.float32, False), (torch.float16, True)])
def test_non_default_weight_dtypes_vs_torch_ref(wdtype, amp):
    """fp32 weights without autocast (amp=False, the reference's fallback when bf16 is unsupported) and fp16 weights
    with fp16 autocast: the W/scale division then happens in fp32 resp. fp16 (torch type promotion)."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("opt", 4, 32, True, seed=41)
    layer = layer.to(wdtype)
    X, others = make_data(rope, cfg, N=8)
    X = X.to(wdtype)

    def tg(blk):
        with torch.no_grad(), torch.autocast("cuda", dtype=wdtype if amp else torch.bfloat16, enabled=amp):
            return torch.cat([fwd(blk, X[i:i + 4], others) for i in range(0, 8, 4)])

    Y = tg(layer)
    blk_o = copy.deepcopy(layer)
    random.seed(8)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=3, batch_size=4, forward=fwd, amp=amp, amp_dtype=wdtype if amp else torch.bfloat16)
    blk_m = copy.deepcopy(layer)
    random.seed(8)
    q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=4, bits=4, amp=amp, amp_dtype=wdtype if amp else torch.bfloat16), device="cuda")
    best_m = q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert abs(st["init_loss"] - info["losses"][0]) <= 2e-3 * info["losses"][0], (st, info["losses"])
    lo, lm = linears(blk_o), linears(blk_m)
    agree = [(lm[n].weight == lo[n].weight).float().mean().item() for n in lo]
    assert all(lm[n].weight.dtype == wdtype for n in lm)
    assert np.mean(agree) > 0.97, agree


@pytest.mark.parametrize("bits,data_type,gs", [(4, "int", 32), (2, "int", 32), (4, "mx_fp", 32), (4, "nv_fp", 16)])
def test_algorithm_extension_block_vs_torch_ref(bits, data_type, gs):
    """SignRoundV2Quantizer.compress_block (imatrix hooks -> searched init scales -> optimized wrapper -> outlier-
    suppressed loss for W2) against torch_ref.tune_block(alg_ext=True), the pinned restatement of the reference's V2."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundV2Quantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("llama", bits, gs, True, seed=2)
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            m.data_type = data_type
            m.act_data_type = data_type
    X, others = make_data(rope, cfg)
    iters, bs = 4, 4

    blk_o = copy.deepcopy(layer)
    Y = targets(blk_o, X, others, bs)
    tr.collect_imatrix(blk_o, X, others, batch_size=bs, forward=fwd)
    im_ref = {n: m.imatrix.clone() for n, m in linears(blk_o).items()}
    wr_init = {}

    def record(i, wrappers, total):
        if i == 0:
            for n, w in wrappers.items():
                wr_init[n] = w.init_scale.reshape(-1).float().clone()

    random.seed(11)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd, alg_ext=True, record=record)

    blk_m = copy.deepcopy(layer)
    random.seed(11)
    q = SignRoundV2Quantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=bits, enable_quanted_input=False), device="cuda")
    # the imatrix the hooks collect == the restatement's
    q.prepare_block(blk_m)
    hs = q.register_fp_input_forward_hooks(blk_m)
    q.forward_all(blk_m, X, others, bs)
    for h in hs:
        h.remove()
    for n, m in linears(blk_m).items():
        assert torch.allclose(m.imatrix, im_ref[n], rtol=1e-5), n
        del m.imatrix
    fp_out, q_out, best_m = q.compress_block(blk_m, X, others)
    assert q._optimized is not None and (q._use_outlier_suppressed_loss == (bits < 4))
    st = q.last_stats
    assert torch.equal(fp_out, Y)
    assert abs(st["init_loss"] - info["losses"][0]) <= 5e-3 * info["losses"][0], (st, info["losses"])
    assert abs(st["best_loss"] - info["best_loss"]) <= 3e-2 * info["best_loss"]
    lo, lm = linears(blk_o), linears(blk_m)
    agree = [(lm[n].weight == lo[n].weight).float().mean().item() for n in lo]
    assert np.mean(agree) > 0.97, agree
    for n in lo:
        assert float(best_m[n]["max_scale"].max()) <= 2.0 and float(best_m[n]["max_scale"].min()) >= 0.0
        assert torch.equal(best_m[n]["min_scale"], torch.ones_like(best_m[n]["min_scale"]))


def test_algorithm_extension_rejects_asym_and_falls_back_like_reference():
    """asym int: the reference keeps the plain WrapperLinear (prepare_run's scheme.sym test); the optimized wrapper itself
    raises for it."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundV2Quantizer
    from auto_round_amd.wrapper import SignRoundOptimizedWrapperLinear, WrapperLinear

    layer, rope, cfg = make_layer("llama", 4, 32, False, seed=2)
    q = SignRoundV2Quantizer(SignRoundConfig(iters=2, batch_size=4), device="cuda")
    q.prepare_block(layer)
    assert not q._optimized and not q._use_outlier_suppressed_loss and q.register_fp_input_forward_hooks(layer) != []
    lin = next(m for m in layer.modules() if isinstance(m, torch.nn.Linear))
    with pytest.raises(ValueError):
        SignRoundOptimizedWrapperLinear(lin, device="cuda")
    assert WrapperLinear.minmax_scale_bound == (0.0, 1.0) and SignRoundOptimizedWrapperLinear.minmax_scale_bound == (0.0, 2.0)


@pytest.mark.parametrize("quanted_input", [False, True])
def test_static_activation_scale_calibration_nvfp4(quanted_input):
    """NVFP4 (nv_fp4_with_static_gs activations): compress_block collects every layer's act_max over ALL calibration
    samples (from the fp-input forward, or from an extra forward on the quantised input when chaining is on), experts
    that saw no token inherit their siblings' maximum, tuning uses that static global scale and packing turns it into
    `input_global_scale` = 448*6/act_max like the reference's llm_compressor / auto_round fp exporters."""
    from auto_round_amd.export import pack_layer
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.testing.moe import build_moe_decoder_layer, set_scheme
    from auto_round_amd.wrapper import WrapperWALayer

    layer, rope, cfg = build_moe_decoder_layer(hidden=256, ffn=512, heads=4, kv_heads=2, num_experts=8, top_k=1, seed=5)
    set_scheme(layer, "NVFP4")
    X, others = make_data(rope.cuda(), cfg, N=2, S=4)      # 8 tokens, top-1 of 8 experts: some experts stay idle
    Xq = (X.float() * 1.5).to(torch.bfloat16)
    src = Xq if quanted_input else X

    # expected maxima: plain hooks on a copy of the fp block
    probe = copy.deepcopy(layer)
    seen = {}

    def mk(name):
        def hook(m, inp, out):
            if inp[0].numel():
                seen[name] = max(seen.get(name, 0.0), float(inp[0].abs().max()))
        return hook

    for n, m in probe.named_modules():
        if isinstance(m, torch.nn.Linear) and getattr(m, "bits", 16) < 16:
            m.register_forward_hook(mk(n))
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        for b0 in range(0, src.shape[0], 2):
            fwd(probe, src[b0:b0 + 2], others)
    idle = [n for n, m in probe.named_modules() if isinstance(m, torch.nn.Linear) and getattr(m, "bits", 16) < 16 and n not in seen]
    assert idle, "the test needs at least one expert without tokens"

    random.seed(3)
    q = SignRoundQuantizer(SignRoundConfig(iters=2, batch_size=2, bits=4, enable_quanted_input=quanted_input), device="cuda")
    q.compress_block(layer, X, others, q_inputs=Xq if quanted_input else None)
    was = {n: m for n, m in layer.named_modules() if isinstance(m, WrapperWALayer)}
    assert len(was) == 4 + 3 * 8
    for n, wa in was.items():
        am = wa.orig_layer.act_max
        assert am.numel() == 1
        if n in seen:
            assert abs(float(am) - seen[n]) <= 1e-6 * seen[n], n
        else:       # idle expert: the maximum over its sibling experts' same-named linear
            leaf = n.rsplit(".", 1)[1]
            sib = max(v for k, v in seen.items() if ".experts." in k and k.endswith("." + leaf))
            assert abs(float(am) - sib) <= 1e-6 * sib, n
    # packing consumes act_max -> input_global_scale
    n0, wa0 = next(iter(was.items()))
    amax = float(wa0.orig_layer.act_max)
    ql = pack_layer(wa0.orig_layer)
    assert abs(float(ql.input_global_scale) - 448.0 * 6.0 / amax) <= 1e-6 * 448.0 * 6.0 / amax
    assert not hasattr(wa0.orig_layer, "act_max") and ql.weight_packed.dtype == torch.uint8


def _dp_worker(rank, world, port, out_path, overlap=True):
    import torch.distributed as dist

    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        layer, rope, cfg = make_layer("llama", 4, 32, True, seed=4)
        X, others = make_data(rope, cfg, N=16, S=16)
        Y = targets(layer, X, others)
        random.seed(21)
        q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=4, bits=4, data_parallel=True, dp_overlap=overlap), device="cuda")
        best = q.quantize_block(layer, X, others, Y, None, None)
        assert q.last_dp_overlapped == overlap          # a dense block: per-layer buckets started from the backward pass
        if rank == 0:
            torch.save({"stats": q.last_stats, "weights": {n: m.weight.cpu() for n, m in linears(layer).items()},
                        "V": {n: b["value"].cpu() for n, b in best.items()}}, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("overlap", [True, False])
def test_data_parallel_block_tuning_two_ranks_one_gpu(tmp_path, overlap):
    """data_parallel=True: two ranks (sharing this box's single GPU, gloo transport) each run half of every minibatch and
    all-reduce the block's dWq buffer once per iteration; the result follows the single-process run (identical iteration-0
    loss up to GEMM batch-split rounding, same trajectory statistics)."""
    import socket

    import torch.multiprocessing as mp

    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "dp.pt")
    mp.spawn(_dp_worker, args=(2, port, out, overlap), nprocs=2, join=True)
    dp = torch.load(out)

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=4)
    X, others = make_data(rope, cfg, N=16, S=16)
    Y = targets(layer, X, others)
    random.seed(21)
    q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=4, bits=4), device="cuda")
    best = q.quantize_block(layer, X, others, Y, None, None)
    st = q.last_stats
    assert abs(dp["stats"]["init_loss"] - st["init_loss"]) <= 5e-3 * st["init_loss"], (dp["stats"], st)
    assert abs(dp["stats"]["best_loss"] - st["best_loss"]) <= 5e-2 * st["best_loss"]
    agree = [(dp["weights"][n] == m.weight.cpu()).float().mean().item() for n, m in linears(layer).items()]
    assert np.mean(agree) > 0.97, agree
    # after the first step V is +-lr0 wherever the summed gradient is non-zero: the two runs agree on almost every sign
    same = [(dp["V"][n].sign() == best[n]["value"].cpu().sign()).float().mean().item() for n in best]
    assert np.mean(same) > 0.9, same


@pytest.mark.parametrize("act_gs,act_sym", [(32, True), (-1, True), (32, False)])
def test_w4a8_int_activation_scheme_vs_torch_ref(act_gs, act_sym):
    """W4A8 (the reference's own GPU smoke configuration, test_sign_sgd_pipeline.py:59-82: bits=4, act_bits=8,
    act_group_size=32, sym): int activation fake-quant inside the tuning loop, WrapperWALayer after unwrapping."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.wrapper import WrapperWALayer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=6)
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            m.act_bits, m.act_group_size, m.act_sym, m.act_dynamic, m.act_data_type = 8, act_gs, act_sym, True, "int"
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 4, 4
    blk_o = copy.deepcopy(layer)
    random.seed(11)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd)
    blk_m = copy.deepcopy(layer)
    random.seed(11)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=4), device="cuda")
    q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert abs(st["init_loss"] - info["losses"][0]) <= 5e-3 * info["losses"][0], (st, info["losses"])
    assert abs(st["best_loss"] - info["best_loss"]) <= 3e-2 * info["best_loss"]
    was = [m for m in blk_m.modules() if isinstance(m, WrapperWALayer)]
    assert len(was) == 7
    agree = [(mo.orig_layer.weight == mm.orig_layer.weight).float().mean().item()
             for mo, mm in zip([m for m in blk_o.modules() if isinstance(m, tr.RefWALayer)], was)]
    assert np.mean(agree) > 0.97, agree
    # the unwrapped block keeps quantising its activations: same output as the restatement's
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.bfloat16):
        a, b = fwd(blk_m, X[:2], others).float(), fwd(blk_o, X[:2], others).float()
    assert float((a - b).abs().mean()) <= 0.05 * float(b.abs().mean())


@pytest.mark.parametrize("family", ["qwen2", "qwen3", "gemma2"])
def test_other_decoder_families_vs_torch_ref(family):
    """The block is a black box for the path (like for the reference): decoder layers of other families -- q/k/v biases
    (Qwen2), q/k norms (Qwen3), pre+post feed-forward norms and soft-capping config (Gemma2) -- tune through the same code and
    agree with the torch restatement."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    torch.manual_seed(0)
    common = dict(hidden_size=256, intermediate_size=512, num_attention_heads=4, num_key_value_heads=2, num_hidden_layers=1,
                  vocab_size=128, max_position_embeddings=128)
    if family == "qwen2":
        from transformers import Qwen2Config as C
        from transformers.models.qwen2.modeling_qwen2 import Qwen2DecoderLayer as L, Qwen2RotaryEmbedding as R
        cfg = C(**common)
    elif family == "qwen3":
        from transformers import Qwen3Config as C
        from transformers.models.qwen3.modeling_qwen3 import Qwen3DecoderLayer as L, Qwen3RotaryEmbedding as R
        cfg = C(head_dim=64, **common)
    else:
        from transformers import Gemma2Config as C
        from transformers.models.gemma2.modeling_gemma2 import Gemma2DecoderLayer as L, Gemma2RotaryEmbedding as R
        cfg = C(head_dim=64, **common)
    cfg._attn_implementation = "sdpa"
    layer = L(cfg, 0).to(torch.bfloat16).eval().cuda()
    rope = R(cfg).cuda()
    n_bias = 0
    for p in layer.parameters():
        p.requires_grad_(False)
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            m.bits, m.group_size, m.sym, m.data_type, m.scale_dtype, m.act_bits = 4, 32, True, "int", torch.float16, 16
            n_bias += m.bias is not None
    assert (n_bias == 3) == (family == "qwen2")
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 4, 4
    blk_o = copy.deepcopy(layer)
    random.seed(5)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd)
    blk_m = copy.deepcopy(layer)
    random.seed(5)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=4, sdpa_backend="auto"), device="cuda")
    q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert st["quantized"] == 7
    assert abs(st["init_loss"] - info["losses"][0]) <= 5e-3 * info["losses"][0], (st, info["losses"])
    lo, lm = linears(blk_o), linears(blk_m)
    agree = [(lm[n].weight == lo[n].weight).float().mean().item() for n in lo]
    assert np.mean(agree) > 0.97, agree
    for n in lo:
        if lo[n].bias is not None:
            assert torch.equal(lm[n].bias, lo[n].bias)      # biases are not tuned (norm/bias tuning is off by default)


@pytest.mark.parametrize("preset,alg_ext", [("INT8", False), ("INT4", True)])
def test_int8_and_int4_presets_per_channel_weights_per_token_activations(preset, alg_ext):
    """The reference's INT8 (W8A8) / INT4 (W4A4) presets: per-output-channel weights (group_size -1) and dynamic per-token
    symmetric int activations; INT4 with the algorithm extension exercises its W-int4/A-int4 special cases (no imatrix
    hooks, outlier-suppressed loss)."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer, SignRoundV2Quantizer
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer("llama", 4, 32, True, seed=8)
    sch = resolve_scheme(preset)
    apply_scheme(layer, sch)
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 3, 4
    blk_o = copy.deepcopy(layer)
    random.seed(2)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd, alg_ext=alg_ext)
    blk_m = copy.deepcopy(layer)
    random.seed(2)
    qcls = SignRoundV2Quantizer if alg_ext else SignRoundQuantizer
    q = qcls(SignRoundConfig(iters=iters, batch_size=bs, bits=sch["bits"], enable_quanted_input=False), device="cuda")
    if alg_ext:
        q.prepare_block(blk_m)
        assert q._is_wint4aint4() and q.register_fp_input_forward_hooks(blk_m) == [] and q._use_outlier_suppressed_loss
        q._scheme = None
    q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert st["quantized"] == 7
    assert abs(st["init_loss"] - info["losses"][0]) <= 1e-2 * info["losses"][0], (st, info["losses"])
    wo = [m.orig_layer for m in blk_o.modules() if isinstance(m, tr.RefWALayer)]
    from auto_round_amd.wrapper import WrapperWALayer
    wm = [m.orig_layer for m in blk_m.modules() if isinstance(m, WrapperWALayer)]
    assert len(wo) == len(wm) == 7
    for a, b in zip(wo, wm):
        assert tuple(b.scale.shape) == (b.weight.shape[0], 1)           # one scale per output channel
    agree = [(a.weight == b.weight).float().mean().item() for a, b in zip(wo, wm)]
    assert np.mean(agree) > 0.95, agree


def _make_stack(n_blocks=4):
    blocks, rope, cfg = [], None, None
    for k in range(n_blocks):
        layer, rope, cfg = make_layer("llama", 4, 32, True, seed=20 + k)
        blocks.append(layer)
    return blocks, rope, cfg


def _shard_worker(rank, world, port, out_path):
    import torch.distributed as dist

    from auto_round_amd import sharding as sh
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    torch.cuda.set_device(0)
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    try:
        blocks, rope, cfg = _make_stack()
        X, others = make_data(rope, cfg, N=8, S=16)
        if rank != 0:
            X = torch.empty_like(X)                     # only rank 0 holds the calibration activations
        q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=4, bits=4, enable_quanted_input=False), device="cuda")
        local = sh.tune_sharded(blocks, X, others, q, seed=42)
        assert sorted(local) == sh.assign_blocks(4, world)[rank]
        payload = {k: {"stats": v["stats"], "weights": {n: m.weight.cpu() for n, m in linears(blocks[k]).items()}}
                   for k, v in local.items()}
        merged = sh.gather_results(payload)
        if rank == 0:
            torch.save(merged, out_path)
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_block_sharding_two_ranks_equals_the_sequential_run(tmp_path):
    """tier brief (5): independent blocks shard across ranks (here 2 ranks sharing the one GPU over gloo): calibration
    broadcast, fp-chain relay and per-block replay of the `random` stream make every block's result IDENTICAL to what a
    sequential single-process run with enable_quanted_input=False produces (same inputs, same index schedule, same kernels)."""
    import socket

    import torch.multiprocessing as mp
    import transformers

    from auto_round_amd.model_tuner import tune_blocks
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    out = str(tmp_path / "sharded.pt")
    mp.spawn(_shard_worker, args=(2, port, out), nprocs=2, join=True)
    sharded = torch.load(out)
    assert sorted(sharded) == [0, 1, 2, 3]

    blocks, rope, cfg = _make_stack()
    X, others = make_data(rope, cfg, N=8, S=16)
    transformers.set_seed(42)                       # what the sequential (reference-style) run does once
    q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=4, bits=4, enable_quanted_input=False), device="cuda")
    recs = tune_blocks(blocks, X, others, q)
    for k in range(4):
        assert abs(sharded[k]["stats"]["init_loss"] - recs[k]["stats"]["init_loss"]) <= 1e-6 * recs[k]["stats"]["init_loss"], k
        for n, m in linears(blocks[k]).items():
            assert torch.equal(sharded[k]["weights"][n], m.weight.cpu()), (k, n)


@pytest.mark.parametrize("kind,bits,gs,sym,scheme", [("llama", 4, 32, True, None), ("llama", 2, 32, False, None), ("llama", 4, 32, True, "MXFP4_W")])
def test_momentum_takes_the_unfused_route_and_tracks_the_torch_restatement(kind, bits, gs, sym, scheme):
    """SignSGD momentum (sign_sgd.py:356-389, off by default): running buffers over materialised gradients."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from oracle import torch_ref as tr

    layer, rope, cfg = make_layer(kind, bits, gs, sym)
    if scheme is not None:
        from auto_round_amd.testing.moe import set_scheme

        set_scheme(layer, scheme)
    X, others = make_data(rope, cfg)
    Y = targets(layer, X, others)
    iters, bs = 6, 4
    blk_o = copy.deepcopy(layer)
    random.seed(13)
    best_o, info = tr.tune_block(blk_o, X, Y, others, iters=iters, batch_size=bs, forward=fwd, momentum=0.9)
    blk_m = copy.deepcopy(layer)
    random.seed(13)
    q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=bs, bits=bits, momentum=0.9), device="cuda")
    q.quantize_block(blk_m, X, others, Y, None, None)
    st = q.last_stats
    assert abs(st["init_loss"] - info["losses"][0]) <= 2e-3 * info["losses"][0], (st, info["losses"])
    assert abs(st["best_loss"] - info["best_loss"]) <= 5e-2 * info["best_loss"], (st, info)
    lo, lm = linears(blk_o), linears(blk_m)
    agree = [(lm[n].weight == lo[n].weight).float().mean().item() for n in lo]
    assert np.mean(agree) > 0.95, agree
