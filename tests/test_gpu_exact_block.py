"""`exact_rounding` on the GPU: csrc/ar_exact.hip against the eager op chains it stands for -- transformers' own modules under torch
autograd ON THE SAME DEVICE, which is what the reference computes when it runs here -- bit for bit, and ExactLlamaBlock against the
module path (block output, every weight gradient, then whole tuned blocks).  The kernels are called through the C ABI (ops.*_exact).
The oracle for this file is torch itself: the reference's arithmetic for these ops IS torch's eager kernels."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
BF = torch.bfloat16


def bits(t):
    return t.contiguous().view({2: torch.int16, 4: torch.int32}[t.element_size()])


def same(a, b):
    return a.shape == b.shape and bool(torch.equal(bits(a), bits(b)))


def gen(seed=0):
    return torch.Generator(device=DEV).manual_seed(seed)


# (rows, hidden): short rows, a non-power-of-two row (Qwen2-7B), Llama-3-8B's minibatch, rows ATen splits over 8 thread-rows (70B)
# ... and 128-value rows (round 6: Qwen3's per-head q / k norms, rows = tokens x heads)
@pytest.mark.parametrize("T,H", [(64, 256), (128, 768), (2048, 3584), (16384, 4096), (4096, 8192), (512, 8320), (16384 * 8, 128), (4096, 128)])
def test_rmsnorm_forward_and_backward_have_eager_torchs_bits(T, H):
    from transformers.models.llama.modeling_llama import LlamaRMSNorm

    from auto_round_amd import ops

    g = gen(T + H)
    norm = LlamaRMSNorm(H, 1e-5).to(BF).to(DEV)
    with torch.no_grad():
        norm.weight.copy_(1 + 0.2 * torch.randn(H, device=DEV, generator=g))
    w = norm.weight.detach()
    x = torch.randn(T, H, device=DEV, generator=g).to(BF)
    r = (0.5 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
    dy = (0.01 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
    dres = (0.01 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
    s = x + r                                                   # the residual add, its own eager op
    sl = s.detach().requires_grad_(True)
    y = norm(sl.view(1, T, H))
    y.backward(dy.view(1, T, H))
    rstd_t = torch.rsqrt(s.float().pow(2).mean(-1, keepdim=True) + 1e-5).view(-1)

    y2, rstd, s2 = ops.rmsnorm_fwd_exact(x, w, 1e-5, res=r)
    assert same(s2, s)
    assert same(ops.rmsnorm_fwd_exact(s, w, 1e-5, raw_sum=True)[1], s.float().pow(2).sum(-1))      # ATen's reduction order
    assert same(rstd, rstd_t)                                    # fp32 row statistics: mean factor, eps, torch's rsqrt
    assert same(y2, y.detach().view(T, H))
    assert same(ops.rmsnorm_bwd_exact(dy, s2, w, rstd), sl.grad)
    assert same(ops.rmsnorm_bwd_exact(dy, s2, w, rstd, dres=dres), sl.grad + dres)


def test_rmsnorm_exact_refuses_what_it_does_not_mirror():
    from auto_round_amd import ops

    w = torch.ones(64, dtype=BF, device=DEV)
    assert ops.rmsnorm_fwd_exact(torch.zeros(64, 64, dtype=BF, device=DEV), w, 1e-5) is None             # short rows: another ATen path
    w = torch.ones(512, dtype=BF, device=DEV)
    assert ops.rmsnorm_fwd_exact(torch.zeros(4, 512, dtype=BF, device=DEV), w, 1e-5) is None             # < 8 rows: a wider thread row


@pytest.mark.parametrize("B,S,hq,hkv,d", [(2, 128, 4, 2, 64), (8, 2048, 32, 8, 128), (1, 256, 8, 8, 96)])
def test_rotary_forward_and_backward_have_eager_torchs_bits(B, S, hq, hkv, d):
    from transformers.models.llama.modeling_llama import apply_rotary_pos_emb

    from auto_round_amd import ops

    if d % 16:
        pytest.skip("head size must be a multiple of 16")
    g = gen(S + d)
    T = B * S
    q = torch.randn(T, hq * d, device=DEV, generator=g).to(BF)
    k = torch.randn(T, hkv * d, device=DEV, generator=g).to(BF)
    ang = torch.rand(1, S, d, device=DEV, generator=g) * 6.28
    cos, sin = ang.cos().to(BF).contiguous(), ang.sin().to(BF).contiguous()
    ql = q.view(B, S, hq, d).transpose(1, 2).detach().requires_grad_(True)
    kl = k.view(B, S, hkv, d).transpose(1, 2).detach().requires_grad_(True)
    qr, kr = apply_rotary_pos_emb(ql, kl, cos, sin)
    gq = torch.randn(B, hq, S, d, device=DEV, generator=g).to(BF)
    gk = torch.randn(B, hkv, S, d, device=DEV, generator=g).to(BF)
    dq, dk = torch.autograd.grad((qr, kr), (ql, kl), (gq, gk))
    tok = lambda t: t.transpose(1, 2).reshape(T, -1)  # noqa: E731

    q2, k2 = ops.rope_fwd_exact(q, k, cos, sin, S, hq, hkv, d)
    assert same(q2, tok(qr.detach())) and same(k2, tok(kr.detach()))
    # column slices of a merged projection as inputs
    qkv = torch.cat([q, k, k], dim=1).contiguous()
    q3, k3 = ops.rope_fwd_exact(qkv[:, :hq * d], qkv[:, hq * d:(hq + hkv) * d], cos, sin, S, hq, hkv, d)
    assert same(q3, q2) and same(k3, k2)
    dq2, dk2 = ops.rope_bwd_exact(gq, gk, cos, sin, S, d)
    assert same(dq2, tok(dq)) and same(dk2, tok(dk))
    # token-major strides (what the flash backward leaves) and a merged gradient buffer as the destination
    merged = torch.empty(T, (hq + 2 * hkv) * d, dtype=BF, device=DEV)
    ops.rope_bwd_exact(gq.transpose(1, 2).contiguous().transpose(1, 2), gk, cos, sin, S, d,
                       out=(merged[:, :hq * d], merged[:, hq * d:(hq + hkv) * d]))
    assert same(merged[:, :hq * d], tok(dq)) and same(merged[:, hq * d:(hq + hkv) * d], tok(dk))


def test_swiglu_forward_is_eager_torch_for_every_bf16_gate_value_and_backward_matches_autograd():
    from auto_round_amd import ops

    g = gen(7)
    allb = torch.arange(65536, device=DEV, dtype=torch.int32).to(torch.int16).view(BF)
    allb = allb[torch.isfinite(allb.float())]
    n = allb.numel() // 8 * 8
    gate = allb[:n].repeat(64, 1).contiguous()
    up = torch.randn(64, n, device=DEV, generator=g).to(BF)
    da = torch.randn(64, n, device=DEV, generator=g).to(BF)
    gl, ul = gate.clone().requires_grad_(True), up.clone().requires_grad_(True)
    a = torch.nn.functional.silu(gl) * ul
    dgl, dul = torch.autograd.grad(a, (gl, ul), da)
    assert same(ops.swiglu_fwd_exact(gate, up), a.detach())
    ok = []
    for contract in (True, False):                      # one of the two forms of silu_backward's inner expression is ATen's
        dg, du = ops.swiglu_bwd_exact(da, gate, up, contract=contract)
        assert same(du, dul)
        ok.append(same(dg, dgl))
    assert any(ok), ok
    # realistic magnitudes, the halves of a merged gradient buffer as the destination, column slices of a merged projection as input
    F_ = 14336
    gu = torch.randn(2048, 2 * F_, device=DEV, generator=g).to(BF)
    da = (0.01 * torch.randn(2048, F_, device=DEV, generator=g)).to(BF)
    gl, ul = gu[:, :F_].clone().requires_grad_(True), gu[:, F_:].clone().requires_grad_(True)
    a = torch.nn.functional.silu(gl) * ul
    dgl, dul = torch.autograd.grad(a, (gl, ul), da)
    assert same(ops.swiglu_fwd_exact(gu[:, :F_], gu[:, F_:]), a.detach())
    dgu = torch.empty_like(gu)
    ops.swiglu_bwd_exact(da, gu[:, :F_], gu[:, F_:], contract=ok[0], out=(dgu[:, :F_], dgu[:, F_:]))
    assert same(dgu[:, :F_], dgl) and same(dgu[:, F_:], dul)


# ---- the block ---------------------------------------------------------------------------------------------------------------
def _small_llama(hidden=512, inter=1024, heads=4, kv=2, seq=256, nsamples=16, layers=1, seed=0):
    from transformers import LlamaConfig, LlamaForCausalLM

    torch.manual_seed(seed)
    cfg = LlamaConfig(hidden_size=hidden, intermediate_size=inter, num_attention_heads=heads, num_key_value_heads=kv,
                      num_hidden_layers=layers, vocab_size=512, max_position_embeddings=1024, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    model = LlamaForCausalLM(cfg).to(BF).eval().to(DEV)
    for p in model.parameters():
        p.requires_grad_(False)
    tokens = torch.randint(0, 512, (nsamples, seq), generator=torch.Generator().manual_seed(1))
    return model, tokens


def _tune(model, tokens, *, exact, with_mask, iters=12, scheme="W4A16", seed=42):
    import transformers

    from auto_round_amd.autoround import loss_mask_ids
    from auto_round_amd.export import pack_block
    from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from auto_round_amd.testing import t3_fixture as fx

    import copy

    model = copy.deepcopy(model)
    block = fx.decoder_blocks(model)[0]
    sch = resolve_scheme(scheme)
    apply_scheme(block, sch)
    x0, others = fx.capture_block_inputs(model, block, tokens, torch.device(DEV))
    if not with_mask:
        others = {k: v for k, v in others.items() if k != "attention_mask"}
    cfg = SignRoundConfig(iters=iters, batch_size=8, bits=sch["bits"], sdpa_backend="auto", exact_rounding=exact)
    q = SignRoundQuantizer(cfg, device=DEV)
    y = q.calibrate_block(block, x0, others)
    transformers.set_seed(seed)
    q.quantize_block(block, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=loss_mask_ids(tokens, None))
    torch.cuda.synchronize()
    packed = {n: tuple(t.clone().view(torch.uint8) if t.dtype == torch.float8_e4m3fn else t.clone() for _, t in sorted(ql.state_dict().items()))
              for n, ql in pack_block(block).items()}
    return q, packed


@pytest.mark.parametrize("with_mask", [True, False])
def test_exact_block_is_proven_against_the_module_path_and_tunes_to_identical_packed_weights(with_mask):
    """A small Llama block (hidden 512, GQA, head size 128) through the whole tuning loop twice: module path and exact_rounding.  The plan
    must have been proven (first-party norm / rotary / SwiGLU kernels in it) and every packed tensor must be identical."""
    model, tokens = _small_llama()
    q_mod, packed_mod = _tune(model, tokens, exact=False, with_mask=with_mask)
    q_ex, packed_ex = _tune(model, tokens, exact=True, with_mask=with_mask)
    assert not q_mod.last_exact and q_ex.last_exact
    rep = q_ex.last_exact_report
    assert rep and rep["usable"], rep
    plan = rep["plan"]
    assert plan["norm1"] and plan["norm2"] and plan["rope"] and plan["swiglu"], rep
    assert q_ex.last_stats["loss_trace"] == q_mod.last_stats["loss_trace"]
    assert sorted(packed_ex) == sorted(packed_mod)
    for n in packed_mod:
        for a, b in zip(packed_ex[n], packed_mod[n]):
            assert torch.equal(a, b), n


def test_attention_joins_the_plan_at_a_sequence_length_where_the_library_uses_another_key_block():
    """seq 512 at head size 128: the library's forward runs 32-key blocks there (64 at the tuning minibatch's 2048) -- the proof finds the
    key block (`attn_kb`: 0 = the measured guess matched) and the attention runs first-party; packed weights equal the module path's"""
    model, tokens = _small_llama(seq=512, nsamples=16)
    q_mod, packed_mod = _tune(model, tokens, exact=False, with_mask=True, iters=6)
    q_ex, packed_ex = _tune(model, tokens, exact=True, with_mask=True, iters=6)
    rep = q_ex.last_exact_report
    assert q_ex.last_exact and rep["usable"] and rep["plan"]["attn"], rep
    assert rep["attn_direct"][-1] == {"out": 0, "dq": 0, "dk": 0, "dv": 0}, rep
    assert q_ex.last_stats["loss_trace"] == q_mod.last_stats["loss_trace"]
    for n in packed_mod:
        for a, b in zip(packed_ex[n], packed_mod[n]):
            assert torch.equal(a, b), n


def test_qwen3_block_with_per_head_norms_runs_on_the_exact_path():
    """Qwen3 (round 6): q_norm / k_norm over every 128-value head run on the RMSNorm kernels with rows = tokens x heads (option
    `qknorm`; ATen reduces a 128-value row as it does a longer one: tools/gpu/r06_headnorm_probe.py) -- the block is proven against the
    module code like a Llama block and tunes to the module path's packed weights"""
    from transformers import Qwen3Config, Qwen3ForCausalLM

    torch.manual_seed(0)
    cfg = Qwen3Config(hidden_size=512, intermediate_size=1024, num_attention_heads=4, num_key_value_heads=2, head_dim=128,
                      num_hidden_layers=1, vocab_size=512, max_position_embeddings=1024, tie_word_embeddings=False)
    cfg._attn_implementation = "sdpa"
    model = Qwen3ForCausalLM(cfg).to(BF).eval().to(DEV)
    for p in model.parameters():
        p.requires_grad_(False)
    with torch.no_grad():       # (non-trivial norm weights)
        for m in model.modules():
            if type(m).__name__ == "Qwen3RMSNorm" and m.weight.numel() == 128:
                m.weight.copy_((1 + 0.1 * torch.randn(128, device=DEV)).to(BF))
    tokens = torch.randint(0, 512, (16, 512), generator=torch.Generator().manual_seed(1))
    q_mod, packed_mod = _tune(model, tokens, exact=False, with_mask=True, iters=6)
    q_ex, packed_ex = _tune(model, tokens, exact=True, with_mask=True, iters=6)
    rep = q_ex.last_exact_report
    assert q_ex.last_exact and rep["usable"], rep
    assert rep["plan"]["qknorm"] and rep["plan"]["norm1"] and rep["plan"]["rope"] and rep["plan"]["attn"], rep
    assert q_ex.last_stats["loss_trace"] == q_mod.last_stats["loss_trace"]
    for n in packed_mod:
        for a, b in zip(packed_ex[n], packed_mod[n]):
            assert torch.equal(a, b), n


def test_exact_plan_is_reused_for_later_blocks_of_the_same_kind_and_refused_for_unsupported_ones():
    from auto_round_amd.exact_block import ExactLlamaBlock
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from auto_round_amd.testing import t3_fixture as fx

    model, tokens = _small_llama(layers=2)
    blocks = fx.decoder_blocks(model)
    sch = resolve_scheme("W4A16")
    for b in blocks:
        apply_scheme(b, sch)
    x0, others = fx.capture_block_inputs(model, blocks[0], tokens, torch.device(DEV))
    q = SignRoundQuantizer(SignRoundConfig(iters=3, batch_size=8, bits=4, sdpa_backend="auto", exact_rounding=True), device=DEV)
    x = x0
    for b in blocks:
        fp_out, q_out, _ = q.compress_block(b, x, others)
        assert q.last_exact
        x = fp_out
    kinds = [k[0] for k in q._exact_plans]
    assert kinds.count("exact") == 1 and kinds.count("exact_plain") == 1        # one proof per form, two blocks
    # a block of no known family is not covered: exact_rounding then means the module path, never the (inexact) fused path
    assert ExactLlamaBlock.try_build(torch.nn.Linear(4, 4), [], {}) is None


def test_no_grad_passes_run_on_the_exact_kernels_and_keep_the_module_codes_bits():
    """composer steps 3 and 6 (targets, quantised-output forward) with exact_rounding: `forward_all` through ExactLlamaBlock's no-grad
    form, proven against the module code, bit-identical to the module path's `forward_all`."""
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.testing import t3_fixture as fx

    model, tokens = _small_llama()
    block = fx.decoder_blocks(model)[0]
    x0, others = fx.capture_block_inputs(model, block, tokens, torch.device(DEV))
    q_mod = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, sdpa_backend="auto"), device=DEV)
    q_ex = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, sdpa_backend="auto", exact_rounding=True), device=DEV)
    y_mod = q_mod.forward_all(block, x0, others)
    y_ex = q_ex.forward_all(block, x0, others)
    plans = [v for k, v in q_ex._exact_plans.items() if k[0] == "exact_plain"]
    assert plans and plans[0] and all(plans[0][k] for k in ("norm1", "norm2", "rope", "swiglu")), plans
    assert same(y_ex, y_mod)
    # a hooked projection (calibration hooks must see the module calls) keeps the module path
    h = block.mlp.down_proj.register_forward_hook(lambda m, i, o: None)
    try:
        assert q_ex._build_exact_plain(block, x0, others, 8) is None
    finally:
        h.remove()


# ---- the stream-K form of the weight-gradient GEMM (ar_gemm_dw_sk, auto_round_amd/streamk.py) ---------------------------------
def _operands(K, M, N, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    dY = (0.01 * torch.randn(K, M, device=DEV, generator=g)).to(torch.bfloat16)
    X = torch.randn(K, N, device=DEV, generator=g).to(torch.bfloat16)
    return dY, X


@pytest.mark.parametrize("K", [2048, 448])          # 448 = 3.5 chunks of 128 k-rows: the zero-completed last chunk
def test_gemm_dw_sk_is_the_two_part_sum_it_says(K):
    from auto_round_amd import ops
    M, N = 512, 768
    dY, X = _operands(K, M, N, 3)
    tiles = (M // 256) * (N // 256)
    one, got = torch.empty(M, N, dtype=torch.bfloat16, device=DEV), torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    assert ops.gemm_dw(dY, X, one, split=False)
    kcut = torch.zeros(tiles, dtype=torch.int32, device=DEV)
    assert ops.gemm_dw_sk(dY, X, got, kcut)                         # no cut anywhere: the one-pass kernel's bits
    assert torch.equal(got.view(torch.int16), one.view(torch.int16))
    half = (-(-K // 128) // 2) * 128                                # where ar_gemm_dw_ex(nsplit=2) puts its slice boundary
    two = torch.empty_like(one)
    assert ops.gemm_dw(dY, X, two, split=2)
    kcut.fill_(half)
    assert ops.gemm_dw_sk(dY, X, got, kcut)
    assert torch.equal(got.view(torch.int16), two.view(torch.int16))
    # a cut per tile, at multiples of 32 k-rows that are not chunk boundaries: part sums in fp32 from sliced operands
    cuts = [32 * (1 + (5 * t) % (K // 32 - 1)) for t in range(tiles)]
    kcut.copy_(torch.tensor(cuts, dtype=torch.int32))
    assert ops.gemm_dw_sk(dY, X, got, kcut)
    ref = torch.empty(M, N, dtype=torch.float32, device=DEV)
    for t, c in enumerate(cuts):
        r0, c0 = (t // (N // 256)) * 256, (t % (N // 256)) * 256
        a, b = dY[:, r0:r0 + 256].float(), X[:, c0:c0 + 256].float()
        ref[r0:r0 + 256, c0:c0 + 256] = a[:c].t() @ b[:c] + a[c:].t() @ b[c:]
    assert (got.float() - ref).abs().max().item() <= 2.0 ** -7 * ref.abs().max().item()
    # entries that are not a multiple of 32 inside (0, K) mean "one pass"
    kcut.copy_(torch.tensor([16, K, -32, K + 64, 48, 0][:tiles], dtype=torch.int32))
    assert ops.gemm_dw_sk(dY, X, got, kcut)
    assert torch.equal(got.view(torch.int16), one.view(torch.int16))
    with pytest.raises(ValueError):
        ops.gemm_dw_sk(dY, X, got, kcut[:-1].contiguous())


def test_streamk_structure_found_on_one_gradient_pair_reproduces_the_library_on_others():
    """Llama-3-8B's gate / up gradient shape, where the library's kernel streams its last tiles over a fixed grid: whatever the
    installed library does there, the structure `find_on_device` settles on (or the one-pass kernel) must give its bits on
    operands it has not seen."""
    from auto_round_amd import ops, streamk
    K, M, N = 16384, 14336, 4096
    dY, X = _operands(K, M, N, 21)
    streamk._found.pop((torch.device(DEV).index or 0, M, N, K), None)
    st = streamk.find_on_device(dY, X)
    lib = torch.mm(dY.t(), X)
    one = torch.empty_like(lib)
    assert ops.gemm_dw(dY, X, one, split=False)
    if st is None:
        assert torch.equal(one.view(torch.int16), lib.view(torch.int16)), "neither one pass nor a stream-K structure reproduces the library"
        return
    structure, kcut = st
    assert structure.two_part_tiles > 0 and structure.n_dp % structure.grid == 0
    for seed in (22, 23):
        dY2, X2 = _operands(K, M, N, seed)
        got = torch.empty_like(lib)
        assert ops.gemm_dw_sk(dY2, X2, got, kcut)
        assert torch.equal(got.view(torch.int16), torch.mm(dY2.t(), X2).view(torch.int16))


# ---- round 6: nn.LayerNorm with torch's bits (csrc/ar_exact_ln.hip) and the exact form of OPT-style blocks -------------------------------
@pytest.mark.parametrize("T,H", [(16384, 768), (256, 768), (4096, 2048), (1000, 1024), (64, 5120), (4096, 12)])
def test_layernorm_forward_and_backward_have_torchs_bits(T, H):
    """What the module path computes for an OPT block's norms under autocast: x.float() -> native_layer_norm (fp32) -> the next linear's
    cast, and autograd's backward of exactly that chain -- against ar_layernorm_fwd_exact / ar_layernorm_bwd_exact, bit for bit,
    including the fp32 row statistics.  OPT-125M's minibatch (16384 x 768: one float4 per thread), rows with several vectors per
    thread, rows shorter than the workgroup, and a row count that is no multiple of anything."""
    from auto_round_amd import ops

    g = gen(T * 7 + H)
    x = (0.7 * torch.randn(T, H, device=DEV, generator=g) + 0.1).to(BF)
    w = (1 + 0.2 * torch.randn(H, device=DEV, generator=g)).to(BF)
    b = (0.1 * torch.randn(H, device=DEV, generator=g)).to(BF)
    dy = (0.01 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
    dres = (0.01 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
    xl = x.detach().requires_grad_(True)
    with torch.autocast("cuda", dtype=BF):
        yf = torch.nn.functional.layer_norm(xl, (H,), w, b, 1e-5)
    assert yf.dtype == torch.float32                      # autocast's fp32 list
    y_ref = yf.to(BF)
    y_ref.backward(dy)
    _, mean_t, rstd_t = torch.ops.aten.native_layer_norm(x.float(), [H], w.float(), b.float(), 1e-5)

    y, mean, rstd = ops.layernorm_fwd_exact(x, w, b, 1e-5)
    assert same(mean, mean_t.view(-1)) and same(rstd, rstd_t.view(-1))
    assert same(y, y_ref.detach())
    assert ops.layernorm_fwd_exact(x, w, b, 1e-5, want_stats=False)[1] is None
    dx = ops.layernorm_bwd_exact(dy, x, w, mean, rstd)
    assert same(dx, xl.grad)
    assert same(ops.layernorm_bwd_exact(dy, x, w, mean, rstd, dres=dres), xl.grad + dres)


def test_layernorm_exact_refuses_what_it_does_not_mirror():
    from auto_round_amd import ops
    from auto_round_amd._lib import Mi355xLibraryError

    x = torch.randn(64, 770, device=DEV).to(BF)           # hidden % 4 != 0: ATen's non-vectorised kernels
    assert ops.layernorm_fwd_exact(x, torch.ones(770, device=DEV, dtype=BF), torch.zeros(770, device=DEV, dtype=BF), 1e-5) is None
    assert not ops.layernorm_bwd_exact_ok(32768, 768) and ops.layernorm_bwd_exact_ok(16384, 768)
    x = torch.randn(32768, 64, device=DEV).to(BF)         # rows >= 32768: ATen's ROCm build runs cuComputeGradInput there
    w = torch.ones(64, device=DEV, dtype=BF)
    y, mean, rstd = ops.layernorm_fwd_exact(x, w, torch.zeros(64, device=DEV, dtype=BF), 1e-5)
    with pytest.raises(Mi355xLibraryError):
        ops.layernorm_bwd_exact(x, x, w, mean, rstd)


def _small_opt(hidden=512, ffn=2048, heads=8, layers=1, seq=256, nsamples=16):
    from transformers import OPTConfig, OPTForCausalLM

    torch.manual_seed(0)
    cfg = OPTConfig(hidden_size=hidden, ffn_dim=ffn, num_attention_heads=heads, num_hidden_layers=layers, vocab_size=512,
                    max_position_embeddings=1024, word_embed_proj_dim=hidden)
    cfg._attn_implementation = "sdpa"
    model = OPTForCausalLM(cfg).to(BF).eval().to(DEV)
    for p in model.parameters():
        p.requires_grad_(False)
    tokens = torch.randint(0, 512, (nsamples, seq), generator=torch.Generator().manual_seed(1))
    return model, tokens


@pytest.mark.parametrize("with_mask,scheme", [(True, "W4A16"), (False, "W4A16"), (True, "MXFP4")])
def test_exact_opt_block_is_proven_against_the_module_path_and_tunes_to_identical_packed_weights(with_mask, scheme):
    """An OPT block (hidden 512, head size 64, biases everywhere, ReLU MLP) through the whole tuning loop twice: module path and
    exact_rounding (auto_round_amd/exact_opt_block.py).  The plan must have been proven with both LayerNorm kernels in it, the loss
    traces and every packed tensor must be identical."""
    model, tokens = _small_opt()
    q_mod, packed_mod = _tune(model, tokens, exact=False, with_mask=with_mask, scheme=scheme)        # (MXFP4: 4-bit activations between the kernels)
    q_ex, packed_ex = _tune(model, tokens, exact=True, with_mask=with_mask, scheme=scheme)
    if scheme == "MXFP4":       # activation-quantised OPT blocks are refused (the module path quantises the norm's fp32 output): the module
        assert not q_ex.last_exact and not q_mod.last_exact          # path runs, quietly, and the results are the module path's
    else:
        assert not q_mod.last_exact and q_ex.last_exact
        rep = q_ex.last_exact_report
        assert rep and rep["usable"], rep
        assert rep["plan"]["ln1"] and rep["plan"]["ln2"], rep
    assert q_ex.last_stats["loss_trace"] == q_mod.last_stats["loss_trace"]
    assert sorted(packed_ex) == sorted(packed_mod)
    for n in packed_mod:
        for a, b in zip(packed_ex[n], packed_mod[n]):
            assert torch.equal(a, b), n


def test_opt_no_grad_passes_run_on_the_exact_form_and_keep_the_module_codes_bits():
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.testing import t3_fixture as fx

    model, tokens = _small_opt()
    block = fx.decoder_blocks(model)[0]
    x0, others = fx.capture_block_inputs(model, block, tokens, torch.device(DEV))
    q_mod = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, sdpa_backend="auto"), device=DEV)
    q_ex = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, sdpa_backend="auto", exact_rounding=True), device=DEV)
    y_mod = q_mod.forward_all(block, x0, others)
    y_ex = q_ex.forward_all(block, x0, others)
    plans = [v for k, v in q_ex._exact_plans.items() if k[0] == "exact_plain"]
    assert plans and plans[0] and plans[0]["ln1"] and plans[0]["ln2"], plans
    assert same(y_ex, y_mod)


def test_a_block_whose_attention_forward_was_caught_differing_is_restored_and_tuned_again(monkeypatch):
    """`verify_attention_forward`: the retry machinery on a tiny OPT block (activation-quantised, so the finished run leaves shells to
    undo): a first attempt whose flag is forced is thrown away -- fp weights, scheme attributes and the `random` stream restored -- and
    the second attempt must give exactly what an unverified run gives."""
    import transformers

    from auto_round_amd import attention
    from auto_round_amd.autoround import loss_mask_ids
    from auto_round_amd.export import pack_block
    from auto_round_amd.quantizer import BlockContext, SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from auto_round_amd.testing import t3_fixture as fx

    import copy

    base, tokens = _small_opt(hidden=256, ffn=512, heads=4, seq=128, nsamples=8)

    def run(verify):
        model = copy.deepcopy(base)
        block = fx.decoder_blocks(model)[0]
        sch = resolve_scheme("MXFP4")
        apply_scheme(block, sch)
        x0, others = fx.capture_block_inputs(model, block, tokens, torch.device(DEV))
        q = SignRoundQuantizer(SignRoundConfig(iters=6, batch_size=4, bits=4, sdpa_backend="auto", verify_attention_forward=verify), device=DEV)
        y = q.calibrate_block(block, x0, others)
        transformers.set_seed(7)
        q.quantize_block(block, x0, others, y, None, BlockContext(0, 1, "0"), input_ids=loss_mask_ids(tokens, None))
        return q, {n: (m.weight_packed.clone(), m.weight_scale.clone().view(torch.uint8)) for n, m in pack_block(block).items()}

    q0, plain = run(False)
    real = attention.verified_sdpa_forward
    calls = {"n": 0}

    import contextlib

    @contextlib.contextmanager
    def forced(flag):
        calls["n"] += 1
        with real(flag):
            yield
        if calls["n"] == 1:
            flag.fill_(True)            # "a pair differed" during the first attempt

    monkeypatch.setattr(attention, "verified_sdpa_forward", forced)
    with pytest.warns(UserWarning, match="verify_attention_forward"):
        q1, verified = run(True)
    assert calls["n"] == 2 and q1.last_stats["attention_forward_retries"] == 1 and not q1.last_stats["attention_forward_unverified"]
    assert q1.last_stats["loss_trace"] == q0.last_stats["loss_trace"]
    assert sorted(verified) == sorted(plain)
    for n in plain:
        assert torch.equal(verified[n][0], plain[n][0]) and torch.equal(verified[n][1], plain[n][1]), n
