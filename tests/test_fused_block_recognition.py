"""Host logic of auto_round_amd/fused_block.py that needs no GPU: which decoder blocks the fused path recognises and which it
leaves to the generic module path (the reference's counterpart, torch.compile(block_forward), takes every block:
auto_round/utils/device.py:112-122), and that nothing is built on a CPU block."""
import pytest
import torch

from auto_round_amd.fused_block import (FusedLlamaBlock, FusedOPTBlock, _is_rmsnorm, _is_silu, _qk_norm, build_fused_block,
                                        build_fused_block_plain, mfma_dw_pays)


def _qwen3(head_dim=64):
    from transformers import Qwen3Config
    from transformers.models.qwen3.modeling_qwen3 import Qwen3DecoderLayer

    cfg = Qwen3Config(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_key_value_heads=1, head_dim=head_dim,
                      num_hidden_layers=1, vocab_size=64, max_position_embeddings=64)
    return Qwen3DecoderLayer(cfg, 0)


def _llama():
    from transformers import LlamaConfig
    from transformers.models.llama.modeling_llama import LlamaDecoderLayer

    cfg = LlamaConfig(hidden_size=128, intermediate_size=256, num_attention_heads=2, num_key_value_heads=1, num_hidden_layers=1,
                      vocab_size=64, max_position_embeddings=64)
    return LlamaDecoderLayer(cfg, 0)


def _opt(**kw):
    from transformers import OPTConfig
    from transformers.models.opt.modeling_opt import OPTDecoderLayer

    cfg = OPTConfig(hidden_size=128, ffn_dim=256, num_attention_heads=2, num_hidden_layers=1, vocab_size=64, max_position_embeddings=64,
                    word_embed_proj_dim=128, **kw)
    return OPTDecoderLayer(cfg).eval()


def test_per_head_qk_norms_are_recognised_only_in_the_qwen3_form():
    blk = _qwen3()
    wq, wk, eps = _qk_norm(blk.self_attn, 64)
    assert wq is blk.self_attn.q_norm.weight and wk is blk.self_attn.k_norm.weight and eps == blk.self_attn.q_norm.variance_epsilon
    assert _qk_norm(_llama().self_attn, 64) is None                       # no norms: plain Llama form
    assert _qk_norm(blk.self_attn, 48) is False                           # a head size the kernel does not take
    blk.self_attn.k_norm = torch.nn.Identity()
    assert _qk_norm(blk.self_attn, 64) is False                           # only one of the two
    blk = _qwen3()
    blk.self_attn.q_norm = torch.nn.LayerNorm(64)
    assert _qk_norm(blk.self_attn, 64) is False                           # another kind of norm
    blk = _qwen3()
    blk.self_attn.k_norm.variance_epsilon = 1e-3
    assert _qk_norm(blk.self_attn, 64) is False                           # two epsilons: one kernel argument


def test_norm_and_activation_predicates():
    blk = _llama()
    assert _is_rmsnorm(blk.input_layernorm) and not _is_rmsnorm(torch.nn.LayerNorm(8)) and not _is_rmsnorm(torch.nn.Identity())
    assert _is_silu(blk.mlp.act_fn) and _is_silu(torch.nn.functional.silu) and not _is_silu(torch.nn.ReLU())


def test_opt_blocks_shape_check():
    blk = _opt()
    n1, n2, attn, proj = FusedOPTBlock._parts(blk)
    assert FusedOPTBlock._shape_ok(blk, n1, n2, attn, *proj) == 64
    post = _opt(do_layer_norm_before=False)                               # OPT-350m's post-norm form
    assert FusedOPTBlock._shape_ok(post, *FusedOPTBlock._parts(post)[:3], *FusedOPTBlock._parts(post)[3]) is None
    gelu = _opt(activation_function="gelu")
    assert FusedOPTBlock._shape_ok(gelu, *FusedOPTBlock._parts(gelu)[:3], *FusedOPTBlock._parts(gelu)[3]) is None
    drop = _opt(dropout=0.1).train()                                      # dropout only matters in training mode
    assert FusedOPTBlock._shape_ok(drop, *FusedOPTBlock._parts(drop)[:3], *FusedOPTBlock._parts(drop)[3]) is None
    assert FusedOPTBlock._shape_ok(drop.eval(), *FusedOPTBlock._parts(drop)[:3], *FusedOPTBlock._parts(drop)[3]) == 64
    assert FusedOPTBlock._parts(_llama()) is None


def test_nothing_is_built_for_cpu_or_unwrapped_blocks():
    """The fused path is a GPU path with no fallback arithmetic of its own: an unwrapped block, a block without arenas, CPU weights
    and blocks of unknown shape all return None and the caller keeps the module path."""
    for blk in (_llama(), _qwen3(), _opt(), torch.nn.Sequential(torch.nn.Linear(4, 4))):
        assert build_fused_block(blk, [], {}, torch.bfloat16) is None
        assert build_fused_block(blk, [object()], {}, torch.bfloat16) is None            # not wrapped: plain nn.Linear projections
        assert build_fused_block_plain(blk.to(torch.bfloat16), {"position_embeddings": (torch.zeros(1, 4, 64), torch.zeros(1, 4, 64))},
                                       torch.bfloat16) is None                             # CPU weights
    assert FusedLlamaBlock.try_build_plain(_llama().to(torch.bfloat16), {}, torch.bfloat16) is None


@pytest.mark.parametrize("M,N,K,want", [(4096, 4096, 16384, True), (768, 768, 16384, True), (4096, 4096, 1024, False),
                                        (4000, 4096, 16384, False), (4096, 1000, 16384, False)])
def test_the_mfma_weight_gradient_gemm_is_used_where_its_tiles_fit(M, N, K, want):
    assert mfma_dw_pays(M, N, K) is want


@pytest.mark.parametrize("M,N,want", [(4163, 4096, [4096, 67]), (4096, 4096, [4096]), (4000, 4096, [4000]), (8300, 4096, [8192, 108]),
                                      (4163, 14336, [4163]), (4163, 28672, [4163]), (300, 4096, [300]), (40000, 4096, [40000]),
                                      (9000, 2048, [8192, 808])])
def test_expert_gemm_calls_are_cut_at_whole_rounds_of_output_tiles(M, N, want, monkeypatch):
    """FusedMoEBlock._mm_rows (host logic; the measurement behind it: profiles/r03_moe_expert_gemm_vs_rows.jsonl): an expert whose row
    count is a little over a whole number of 256-CU rounds of 256 x 256 output tiles goes in two calls where the output is narrow
    (<= 16 column tiles) and only a few rounds deep; everything else is one call.  The result is the same product either way."""
    from auto_round_amd.fused_block import FusedMoEBlock

    calls = []
    real = torch.mm

    def spy(a, w, out=None):
        calls.append(a.shape[0])
        return real(a, w, out=out)

    monkeypatch.setattr(torch, "mm", spy)
    a = torch.randn(M, 8)
    w = torch.randn(8, N)
    out = torch.empty(M, N)
    FusedMoEBlock._mm_rows(a, w, out)
    assert calls == want
    monkeypatch.undo()
    assert torch.equal(out, a @ w)


def test_attention_backward_refuses_shapes_outside_the_kernel_before_touching_the_library():
    from auto_round_amd import ops

    t = torch.zeros(256, 128, dtype=torch.bfloat16)
    lse = torch.zeros(1, 1, 256)
    assert ops.attn_bwd(t, t, t, t, lse, t, 1, 256, 1, 128) is None                      # head size 128
    assert ops.attn_bwd(t, t, t, t, lse, t, 1, 128, 2, 64) is None                       # sequence not a multiple of 256
    assert ops.attn_bwd(t.float(), t, t, t, lse, t, 1, 256, 2, 64) is None               # not bf16
    with pytest.raises(Exception):                                                        # right shape, CPU tensors: loud, no fallback
        ops.attn_bwd(t, t, t, t, lse, t, 1, 256, 2, 64)


def test_the_captured_graph_form_is_chosen_only_where_the_loop_needs_nothing_from_the_host():
    """SignRoundQuantizer._graph_eligible (host logic): automatic mode = small fused blocks only (measured: at OPT-125M's 7.1 M weights
    the host-driven loop is the faster form); `hip_graph=True` lifts the size limit, never the structural conditions."""
    from types import SimpleNamespace as NS

    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer

    elig = SignRoundQuantizer._graph_eligible
    fused, small, opt, big = NS(capturable=True), [NS(n=1 << 20, shared=False)], [NS(n=7077888, shared=False)], [NS(n=218103808, shared=False)]
    base = dict(early_stop=False, dp_size=1, accum=False, per_sample_others=False, valid_counts=[2047] * 4, sched_dev=object(), track_best=True)
    auto, forced, off = SignRoundConfig(iters=200), SignRoundConfig(iters=200, hip_graph=True), SignRoundConfig(iters=200, hip_graph=False)
    assert elig(auto, fused, small, **base) and not elig(auto, fused, opt, **base) and not elig(auto, fused, big, **base)
    assert elig(forced, fused, opt, **base) and elig(forced, fused, big, **base)
    assert not elig(off, fused, small, **base)
    assert not elig(forced, None, small, **base)                                        # module path: autograd runs the block
    assert not elig(forced, NS(capturable=False), small, **base)                        # sparse MoE: expert counts are read on the host
    for k, v in dict(early_stop=True, dp_size=2, accum=True, per_sample_others=True, sched_dev=None, track_best=False,
                     valid_counts=[2047, 2040]).items():
        assert not elig(forced, fused, small, **{**base, k: v}), k
    assert not elig(SignRoundConfig(iters=200, hip_graph=True, momentum=0.9), fused, small, **base)
    assert not elig(forced, fused, [NS(n=1 << 20, shared=True)], **base)
    assert not elig(SignRoundConfig(iters=2, hip_graph=True), fused, small, **base)
