"""(test infrastructure) The block-by-block driver flow restated once, with oracle/torch_ref as the tuning engine:

    capture block-0 inputs -> per block: [imatrix hooks] fp forward -> [act_max hooks on the quantised input, MoE fill,
    NVFP4 global-scale unification] -> tune -> quantised-output forward -> chain (fp chain + quantised chain)

`tests/test_pipeline_vs_reference.py` pins it bit for bit against the reference's `AutoRound(...).quantize()` on CPU;
`tests/test_gpu_autoround.py` runs it on the GPU next to the product's front door (HIP engine)."""
import torch

from oracle import torch_ref as tr


def fwd(blk, x, others):
    n = others.get("_materialise_rows")
    if n:        # reference-mask mode: every shared tensor is materialised at the batch's own row count (see run_flow)
        rows = x.shape[0]
        others = {k: (tuple(t.expand(rows, *t.shape[1:]).contiguous() for t in v) if isinstance(v, tuple) else
                      (v.expand(rows, *v.shape[1:]).contiguous() if isinstance(v, torch.Tensor) and v.dim() and v.shape[0] == 1 else v))
                  for k, v in others.items() if k != "_materialise_rows"}
    out = blk(x, **others)
    return out[0] if isinstance(out, (tuple, list)) else out


def reference_cached_mask(S, device="cpu"):
    """What the reference's input cache holds with transformers >= 5: the boolean mask (causal AND key != last token, which
    its calibrator masks) cast to bf16 (calibration/inputs.py:100-107)."""
    m = torch.tril(torch.ones(S, S, device=device))
    m[:, -1] = 0
    return m.to(torch.bfloat16).reshape(1, 1, S, S)


def run_flow(model, blocks, tokens, scheme, *, iters, bs, alg_ext=False, moe=False, reference_mask=False, seed=42,
             quanted_input=True, tune_kw=None, pad_token_id=None):
    """Tunes `blocks` of `model` in place (scheme attributes must already be on the linears).  Returns per-block
    (init_loss, best_loss) and the number of MoE layers whose act_max had to be filled in."""
    import transformers

    from auto_round_amd.quantizer import register_act_max_hooks, set_amax_for_uncalibrated_experts
    from auto_round_amd.wrapper import update_block_global_scale_if_needed

    device = next(model.parameters()).device
    S = tokens.shape[1]
    captured, others = [], {}

    class Stop(Exception):
        pass

    def hook(mod, args, kwargs):
        captured.append(args[0].detach())
        if not others:
            for k, v in kwargs.items():
                if k not in ("hidden_states", "past_key_values", "use_cache", "cache_position"):
                    others[k] = tuple(x[:1] for x in v) if isinstance(v, tuple) else (
                        v[:1] if isinstance(v, torch.Tensor) and v.dim() and v.shape[0] == bs else v)
        raise Stop

    h = blocks[0].register_forward_pre_hook(hook, with_kwargs=True)
    with torch.no_grad():
        for b0 in range(0, tokens.shape[0], bs):
            try:
                model(input_ids=tokens[b0:b0 + bs].to(device), use_cache=False)
            except Stop:
                pass
    h.remove()
    x0 = torch.cat(captured, 0)
    if reference_mask:
        others["attention_mask"] = reference_cached_mask(S, device)
        # the reference concatenates its per-sample cache entries into tensors with one row per sample of the batch; a broadcast
        # [1, ...] mask takes another CPU SDPA path whose last-bit differences are enough to move the algorithm extension's
        # importance matrix -> fwd() materialises the shared tensors per call (ragged last batches included)
        others["_materialise_rows"] = True
    from auto_round_amd.autoround import loss_mask_ids

    ids = loss_mask_ids(tokens, pad_token_id)       # the product's own rule for which positions enter the loss

    @torch.no_grad()
    def forward_all(blk, x):
        outs = []
        for b0 in range(0, x.shape[0], bs):
            with torch.autocast(device.type, dtype=torch.bfloat16):
                outs.append(fwd(blk, x[b0:b0 + bs], others))
        return torch.cat(outs, 0)

    transformers.set_seed(seed)
    stats, n_filled = [], 0
    fp_in, q_in = x0, None
    tr.RefWALayer.follow_reference_ignored_act_max = bool(reference_mask)      # see the note on RefWALayer
    try:
        return _run_blocks(blocks, fp_in, q_in, others, scheme, iters, bs, alg_ext, moe, quanted_input, tune_kw, ids, forward_all,
                           stats, n_filled)
    finally:
        tr.RefWALayer.follow_reference_ignored_act_max = False


def _run_blocks(blocks, fp_in, q_in, others, scheme, iters, bs, alg_ext, moe, quanted_input, tune_kw, ids, forward_all, stats, n_filled):
    from auto_round_amd.quantizer import register_act_max_hooks, set_amax_for_uncalibrated_experts
    from auto_round_amd.wrapper import update_block_global_scale_if_needed

    for blk in blocks:
        if alg_ext:          # the imatrix hooks fire during the reference (fp-input) forward
            tr.collect_imatrix(blk, fp_in, others, batch_size=bs, forward=fwd)
        fp_out = forward_all(blk, fp_in)
        xin = q_in if (q_in is not None and quanted_input) else fp_in
        if str(scheme.get("act_data_type", "")).startswith("nv_fp"):   # static activation scales + unified weight global scales
            hooks = register_act_max_hooks(blk)                 # composer.py:430-436: collected on the quantised-input forward
            forward_all(blk, xin)
            for h2 in hooks:
                h2.remove()
            if moe:          # experts that saw no calibration token inherit their siblings' maximum
                n_filled += set_amax_for_uncalibrated_experts(blk)
            update_block_global_scale_if_needed(blk)
        _, info = tr.tune_block(blk, xin, fp_out, others, iters=iters, batch_size=bs, forward=fwd, input_ids=ids, alg_ext=alg_ext,
                                **(tune_kw or {}))
        stats.append((info["losses"][0], info["best_loss"]))
        if quanted_input:        # composer.py:476-481: the next block is tuned on this block's quantised output
            q_in = forward_all(blk, xin)
        fp_in = fp_out
    return stats, n_filled
