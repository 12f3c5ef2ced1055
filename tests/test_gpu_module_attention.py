"""`exact_attention` (round 6): a block that has NO exact form (Qwen3 at head size 64: per-head q / k norms over 64-value rows, which the
exact RMSNorm kernels do not restate) is tuned on the module path with its
attention on csrc/ar_attn_exact.hip -- installed through transformers' AttentionInterface only after the quantizer proved, on two
real minibatches, that block output, weight gradients and every attention call's output / q, k, v gradients equal the stock module
path's (reference: auto_round/compressors/utils.py:109-172 `block_forward` around the model's own attention,
transformers/integrations/sdpa_attention.py).  The tuned, packed weights must be identical with the switch on and off."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV, BF = "cuda", torch.bfloat16


def _qwen3(hidden=768, inter=1024, heads=12, kv=4, head_dim=64, seq=2048, nsamples=16, seed=0):
    from transformers import Qwen3Config, Qwen3ForCausalLM

    torch.manual_seed(seed)
    cfg = Qwen3Config(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_key_value_heads=kv, head_dim=head_dim,
                      num_hidden_layers=1, vocab_size=512, max_position_embeddings=4096, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    model = Qwen3ForCausalLM(cfg).to(BF).eval().to(DEV)
    for p in model.parameters():
        p.requires_grad_(False)
    tokens = torch.randint(0, 512, (nsamples, seq), generator=torch.Generator().manual_seed(1))
    return model, tokens


def _tune(model, tokens, *, exact_attention, iters=6, scheme="W4A16", seed=42):
    import transformers

    from auto_round_amd.autoround import loss_mask_ids
    from auto_round_amd.export import pack_block
    from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from auto_round_amd.testing import t3_fixture as fx

    model = copy.deepcopy(model)
    block = fx.decoder_blocks(model)[0]
    sch = resolve_scheme(scheme)
    apply_scheme(block, sch)
    x0, others = fx.capture_block_inputs(model, block, tokens, torch.device(DEV))
    cfg = SignRoundConfig(iters=iters, batch_size=8, bits=sch["bits"], sdpa_backend="auto", exact_rounding=True, exact_attention=exact_attention)
    q = SignRoundQuantizer(cfg, device=DEV)
    y = q.calibrate_block(block, x0, others)
    transformers.set_seed(seed)
    q.quantize_block(block, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=loss_mask_ids(tokens, None))
    torch.cuda.synchronize()
    impl = {getattr(m.config, "_attn_implementation", None) for m in block.modules() if hasattr(m, "config")}
    packed = {n: tuple(t.clone() for _, t in sorted(ql.state_dict().items())) for n, ql in pack_block(block).items()}
    return q, packed, impl


def test_module_path_block_runs_its_attention_first_party_after_the_proof_and_tunes_to_identical_packed_weights():
    model, tokens = _qwen3()
    q_on, packed_on, impl_on = _tune(model, tokens, exact_attention=True)
    q_off, packed_off, impl_off = _tune(model, tokens, exact_attention=False)
    assert not q_on.last_exact and not q_off.last_exact                        # no exact form for this kind of block: module path
    rep = q_on.last_module_attention_report
    assert q_on.last_module_exact_attention, rep
    assert rep["usable"] and rep["block_mismatches"] == 0 and rep["fallbacks"] == 0 and rep["calls"] >= 2, rep
    assert rep["attn_direct"] == {"out": 0, "dq": 0, "dk": 0, "dv": 0}, rep
    assert not q_off.last_module_exact_attention
    assert impl_on == {"sdpa"} and impl_off == {"sdpa"}                        # the stock attention function is put back after the block
    assert q_on.last_stats["loss_trace"] == q_off.last_stats["loss_trace"]
    assert sorted(packed_on) == sorted(packed_off)
    for n in packed_off:
        for a, b in zip(packed_on[n], packed_off[n]):
            assert torch.equal(a, b), n


def test_a_call_the_kernels_do_not_take_keeps_the_stock_attention():
    """seq 192: the kernels take S % 128 == 0 only -- every call falls through to transformers' own attention, the proof says so and
    nothing is installed"""
    model, tokens = _qwen3(seq=192)
    with pytest.warns(UserWarning, match="exact_attention"):
        q, _, impl = _tune(model, tokens, exact_attention=True, iters=2)
    assert not q.last_exact
    assert not q.last_module_exact_attention and not q.last_module_attention_report["usable"]
    assert q.last_module_attention_report["calls"] == 0 and q.last_module_attention_report["fallbacks"] > 0
    assert impl == {"sdpa"}
