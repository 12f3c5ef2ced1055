"""T3: the REAL reference running on the MI355X is the oracle.

For one tiny model and one scheme the reference's own front door `AutoRound(...).quantize()` runs twice on cuda:0:
  (ref)  with the reference's SignRound quantizer (torch eager on the GPU),
  (hip)  with `auto_round_amd.plugin.register()` + `alg_configs=MI355XSignRoundConfig(...)`: the same orchestrator, calibration
         cache and chaining, but every block tuned by this repository's HIP engine behind the plugin boundary.
Compared per block: the iteration-0 loss (identical fake-quant weights -> identical prediction -> equal loss), the best loss, the
fraction of identical tuned weights / integer codes, and scale / zero-point equality where the codes agree.

`run_case(...)` returns the numbers; tests/test_gpu_t3_reference.py asserts on them; `python tests/t3_compare.py --out f.json`
writes the report committed as profiles/archive/r02_t3_reference_on_mi355x.json.  Needs the reference tree (tests/ref_tree.py)."""
import copy
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from ref_tree import import_reference, reference_root  # noqa: E402

CASES = {
    "llama_w4g32": dict(arch="llama", kw=dict(scheme="W4A16", group_size=32)),
    "llama_w2g32_asym": dict(arch="llama", kw=dict(scheme="W2A16G32", sym=False)),
    "llama_mxfp4": dict(arch="llama", kw=dict(scheme="MXFP4")),
    "llama_w4a8": dict(arch="llama", kw=dict(scheme="W4A16", group_size=32, act_bits=8)),
    "opt_w4g32": dict(arch="opt", kw=dict(scheme="W4A16", group_size=32)),
    "mixtral_w4g32": dict(arch="mixtral", kw=dict(scheme="W4A16", group_size=32)),
    "llama_w2g32_alg_ext": dict(arch="llama", kw=dict(scheme="W2A16G32", enable_alg_ext=True)),
    "llama_w4g32_fp_chain": dict(arch="llama", kw=dict(scheme="W4A16", group_size=32, enable_quanted_input=False)),
}


def _model(arch):
    from test_pipeline_vs_reference import _tiny, _tiny_moe, _tiny_opt

    return {"llama": _tiny, "opt": _tiny_opt, "mixtral": _tiny_moe}[arch]()


def _layers(model):
    from test_pipeline_vs_reference import _layers as L

    return L(model)


class _HipLossTrace:
    """Per-iteration loss of the HIP engine: `ops.best_loss_update(total_loss, ...)` sees every iteration's accumulated loss on
    the device; the probe copies it out (one sync per iteration -- test only)."""

    def __init__(self):
        self.blocks = []

    def install(self):
        import auto_round_amd.ops as ops
        import auto_round_amd.quantizer as product

        self._ops, self._orig = ops, ops.best_loss_update
        trace = self

        def spy(total_loss, state, istate, it, *a, **k):
            if it == 0:
                trace.blocks.append([])
            trace.blocks[-1].append(float(total_loss.item()))
            return trace._orig(total_loss, state, istate, it, *a, **k)

        ops.best_loss_update = spy
        return self

    def remove(self):
        self._ops.best_loss_update = self._orig


class _LossProbe:
    """Records, per quantize_block call of the reference's quantizer classes, the first minibatch loss divided by the
    valid-token count -- the reference's `init_loss` (sign_round/quantizer.py:477-497), which it only logs with 6 decimals."""

    def __init__(self):
        self.init_losses, self._armed, self._num = [], False, 1
        self.traces = []            # per block: every minibatch loss / valid-token count, in order
        self._undo, self._depth = [], 0

    def install(self):
        from auto_round.algorithms.quantization.sign_round.quantizer import SignRoundQuantizer as R
        from auto_round.algorithms.quantization.sign_roundv2.quantizer import SignRoundV2Quantizer as R2

        probe = self

        def wrap_qb(cls):
            orig = cls.__dict__.get("quantize_block")
            if orig is None:
                return

            def quantize_block(self, *a, **k):
                probe._armed, probe._num = True, 1
                probe.traces.append([])
                return orig(self, *a, **k)

            cls.quantize_block = quantize_block
            self._undo.append((cls, "quantize_block", orig))

        def wrap_loss(cls):
            orig = cls.__dict__.get("_get_loss")
            if orig is None:
                return

            def _get_loss(self, *a, **k):
                probe._depth += 1                       # the V2 class defers to the base class: record the outermost call only
                try:
                    loss = orig(self, *a, **k)
                finally:
                    probe._depth -= 1
                if probe._depth == 0:
                    val = float(loss.item()) / max(probe._num, 1)
                    if probe._armed:
                        probe._armed = False
                        probe.init_losses.append(val)
                    if probe.traces:
                        probe.traces[-1].append(val)
                return loss

            cls._get_loss = _get_loss
            self._undo.append((cls, "_get_loss", orig))

        orig_cnt = R._get_non_zero_cnt

        def _get_non_zero_cnt(self, tensor, indices):
            n = orig_cnt(self, tensor, indices)
            probe._num = n
            return n

        R._get_non_zero_cnt = _get_non_zero_cnt
        self._undo.append((R, "_get_non_zero_cnt", orig_cnt))
        for c in (R, R2):
            wrap_qb(c)
            wrap_loss(c)
        return self

    def remove(self):
        for cls, name, orig in reversed(self._undo):
            setattr(cls, name, orig)
        self._undo = []


def _decode(lin):
    """(integer codes, scale fp32, zp fp32) of a tuned INT layer from its baked weight and scale / zp attributes."""
    W = lin.weight.detach().float().cpu()
    from transformers.pytorch_utils import Conv1D

    if isinstance(lin, Conv1D):
        W = W.t()
    out_f, in_f = W.shape
    s = lin.scale.float().reshape(out_f, -1)
    gs = in_f // s.shape[1]
    zp = lin.zp if isinstance(lin.zp, torch.Tensor) else torch.full_like(s, float(lin.zp))
    zp = zp.float().reshape(out_f, -1)
    q = torch.round(W.reshape(out_f, -1, gs) / s.unsqueeze(-1)) + zp.unsqueeze(-1)
    return q, s, zp


def run_case(name, iters=20, nsamples=16, seqlen=32, batch_size=4, seed=42, device_map=0):
    import_reference()
    from auto_round import AutoRound
    from test_pipeline_vs_reference import _Loader, _StubTokenizer

    import auto_round_amd.plugin as plugin
    import auto_round_amd.quantizer as product

    case = CASES[name]
    kw = dict(case["kw"])
    Cfg, _ = plugin.register()
    base = _model(case["arch"])
    tokens = torch.randint(0, 64, (nsamples, seqlen), generator=torch.Generator().manual_seed(1))
    cwd = os.getcwd()
    import tempfile

    work = tempfile.mkdtemp(prefix="t3_")
    os.chdir(work)                       # the reference writes ./ar_work_space
    try:
        common = dict(tokenizer=_StubTokenizer(), nsamples=nsamples, seqlen=seqlen, dataset=_Loader(tokens), device_map=device_map,
                      batch_size=batch_size, enable_torch_compile=False, seed=seed, **kw)
        probe = _LossProbe().install()
        try:
            q_ref, _ = AutoRound(copy.deepcopy(base), iters=iters, **common).quantize()
        finally:
            probe.remove()
        hip_stats = []
        hip_trace = _HipLossTrace().install()
        orig_qb = product.SignRoundQuantizer.quantize_block

        def spy(self, *a, **k):
            out = orig_qb(self, *a, **k)
            hip_stats.append(dict(self.last_stats))
            return out

        product.SignRoundQuantizer.quantize_block = spy
        try:
            q_hip, _ = AutoRound(copy.deepcopy(base), alg_configs=Cfg(iters=iters), **common).quantize()
        finally:
            product.SignRoundQuantizer.quantize_block = orig_qb
            hip_trace.remove()
    finally:
        os.chdir(cwd)

    from transformers.pytorch_utils import Conv1D

    def lin(m):
        out = {}
        for n, p in _layers(m).named_modules():
            if isinstance(p, (torch.nn.Linear, Conv1D)) and hasattr(p, "scale"):
                out[n.replace(".orig_layer", "")] = p
        return out

    Lr, Lh = lin(q_ref), lin(q_hip)
    rec = {"case": name, "scheme": kw, "arch": case["arch"], "iters": iters, "nsamples": nsamples, "seqlen": seqlen,
           "batch_size": batch_size, "layers": len(Lr), "same_layer_set": sorted(Lr) == sorted(Lh),
           "init_loss_ref": probe.init_losses, "init_loss_hip": [s["init_loss"] for s in hip_stats],
           "best_loss_hip": [s["best_loss"] for s in hip_stats], "engine_calls": len(hip_stats)}
    tot = same_w = same_q = same_sz = n_sz = 0
    int_scheme = str(next(iter(Lr.values())).data_type).startswith("int")
    for n, a in Lr.items():
        b = Lh[n]
        wa, wb = a.weight.detach().cpu().view(torch.int16), b.weight.detach().cpu().view(torch.int16)
        tot += wa.numel()
        same_w += int((wa == wb).sum())
        sa, sb = a.scale.float().cpu().reshape(-1), b.scale.float().cpu().reshape(-1)
        if int_scheme:
            qa, s1, z1 = _decode(a)
            qb, s2, z2 = _decode(b)
            eq = (qa == qb)
            same_q += int(eq.sum())
            groups_eq = eq.all(dim=-1)                      # groups whose integer codes all agree
            n_sz += int(groups_eq.sum())
            same_sz += int(((s1 == s2) & (z1 == z2))[groups_eq].sum())
        else:
            n_sz += sa.numel()
            same_sz += int((sa == sb).sum())
    rec["weights"] = tot
    rec["frac_identical_weights"] = same_w / tot
    if int_scheme:
        rec["frac_identical_int_codes"] = same_q / tot
    rec["groups_compared_for_scale_zp"] = n_sz
    rec["frac_identical_scale_zp_where_codes_agree"] = (same_sz / n_sz) if n_sz else None
    rel = [abs(a - b) / max(abs(a), 1e-30) for a, b in zip(rec["init_loss_ref"], rec["init_loss_hip"])]
    rec["init_loss_max_rel_diff"] = max(rel) if rel else None
    # where do the two engines' loss trajectories part?  (iteration of the first relative difference > 1e-4, per block; None =
    # never within the run).  Immediate = a systematic difference; late = sign-SGD chaos seeded by a near-zero gradient.
    first = []
    for tr_ref, tr_hip in zip(probe.traces, hip_trace.blocks):
        n = min(len(tr_ref), len(tr_hip))
        f = next((i for i in range(n) if abs(tr_ref[i] - tr_hip[i]) > 1e-4 * max(abs(tr_ref[i]), 1e-30)), None)
        first.append(f)
    rec["loss_trajectory_first_divergence_iter"] = first
    rec["loss_trajectory_len"] = [min(len(a), len(b)) for a, b in zip(probe.traces, hip_trace.blocks)]
    return rec


def _sharded_worker(rank, world, port, out_path, iters, nsamples, seqlen, batch_size):
    """One of `world` ranks sharing cuda:0 over gloo: run the REFERENCE front door (fp-chain mode) to get its tuned weights and
    the exact block-0 inputs it hands to quantize_block, then tune the same blocks with `sharding.tune_sharded` (HIP engine,
    broadcast + pipelined relay + per-block schedule replay) and compare on rank 0."""
    import torch.distributed as dist

    torch.cuda.set_device(0)
    joined = False
    try:
        import_reference()
        from auto_round import AutoRound
        from auto_round.algorithms.quantization.sign_round.quantizer import SignRoundQuantizer as R
        from test_pipeline_vs_reference import _Loader, _StubTokenizer

        from auto_round_amd import sharding as sh
        from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer, normalize_input_others, stack_samples
        from auto_round_amd.schemes import apply_scheme, resolve_scheme

        base = _model("llama")
        tokens = torch.randint(0, 64, (nsamples, seqlen), generator=torch.Generator().manual_seed(1))
        captured = {}
        orig = R.quantize_block

        def spy(self, block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids=None, **kw):
            if "x0" not in captured:
                captured.update(x0=[t.detach().clone() for t in fp_inputs], others=input_others, ids=input_ids)
            return orig(self, block, fp_inputs, input_others, fp_outputs, q_inputs, block_ctx, input_ids=input_ids, **kw)

        import tempfile

        cwd = os.getcwd()
        os.chdir(tempfile.mkdtemp(prefix=f"t3s{rank}_"))
        R.quantize_block = spy
        try:
            q_ref, _ = AutoRound(copy.deepcopy(base), iters=iters, tokenizer=_StubTokenizer(), nsamples=nsamples, seqlen=seqlen,
                                 dataset=_Loader(tokens), device_map=0, batch_size=batch_size, enable_torch_compile=False, seed=42,
                                 scheme="W4A16", group_size=32, enable_quanted_input=False).quantize()
        finally:
            R.quantize_block = orig
            os.chdir(cwd)
        # the reference switches to its experimental DDP mode as soon as torch.distributed is initialised
        # (utils/distributed.py): it must run BEFORE this process joins the group
        dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
        joined = True
        m = copy.deepcopy(base).to("cuda")
        for p in m.parameters():
            p.requires_grad_(False)
        blocks = list(m.model.layers)
        sch = resolve_scheme(scheme="W4A16", group_size=32)
        for b in blocks:
            apply_scheme(b, sch)
        X = stack_samples(captured["x0"], "cuda")
        if rank != 0:
            X = torch.zeros_like(X)                       # only rank 0 holds the calibration activations; the broadcast delivers them
        shared, per_sample = normalize_input_others(captured["others"], nsamples, "cuda")
        assert not per_sample
        q = SignRoundQuantizer(SignRoundConfig(iters=iters, batch_size=batch_size, bits=4, enable_quanted_input=False), device="cuda")
        mine = [b if sh.owner_of(k, len(blocks), world) == rank else None for k, b in enumerate(blocks)]
        local = sh.tune_sharded(mine, X, shared, q, seed=42, input_ids=captured["ids"])
        payload = {k: {"stats": v["stats"], "weights": {n: mod.weight.detach().cpu() for n, mod in blocks[k].named_modules()
                                                         if isinstance(mod, torch.nn.Linear)}} for k, v in local.items()}
        merged = sh.gather_results(payload)
        if rank == 0:
            tot = same = 0
            for k, rec in merged.items():
                ref_lin = {n: mod for n, mod in q_ref.model.layers[k].named_modules() if isinstance(mod, torch.nn.Linear)}
                for n, w in rec["weights"].items():
                    a, b = ref_lin[n].weight.detach().cpu().view(torch.int16), w.view(torch.int16)
                    tot += a.numel()
                    same += int((a == b).sum())
            out = {"case": "llama_w4g32_fp_chain_sharded_2_ranks", "ranks": world, "blocks": sorted(merged), "weights": tot,
                   "frac_identical_weights": same / tot, "init_loss_hip": [merged[k]["stats"]["init_loss"] for k in sorted(merged)],
                   "owners": {k: sh.owner_of(k, len(blocks), world) for k in sorted(merged)}}
            with open(out_path, "w") as f:
                json.dump(out, f)
        dist.barrier()
    finally:
        if joined:
            dist.destroy_process_group()


def run_sharded_case(iters=20, nsamples=16, seqlen=32, batch_size=4, world=2):
    import socket
    import tempfile

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = os.path.join(tempfile.mkdtemp(prefix="t3s_"), "sharded.json")
    mp.spawn(_sharded_worker, args=(world, port, out, iters, nsamples, seqlen, batch_size), nprocs=world, join=True)
    with open(out) as f:
        return json.load(f)


def main():
    import argparse

    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--iters", type=int, default=20)
    args = ap.parse_args()
    if reference_root() is None:
        raise SystemExit("reference tree not present: run tools/stage_reference.sh first")
    recs = []
    for c in args.cases.split(","):
        try:
            r = run_case(c, iters=args.iters)
        except Exception as e:  # keep going: the report names what failed
            import traceback

            r = {"case": c, "error": repr(e), "trace": traceback.format_exc()[-1500:]}
        recs.append(r)
        print(json.dumps(r), flush=True)
    try:
        r = run_sharded_case(iters=args.iters)
    except Exception as e:
        import traceback

        r = {"case": "llama_w4g32_fp_chain_sharded_2_ranks", "error": repr(e), "trace": traceback.format_exc()[-1500:]}
    recs.append(r)
    print(json.dumps(r), flush=True)
    if args.out:
        with open(args.out, "w") as f:
            json.dump({"what": "reference AutoRound(...).quantize() on cuda:0 vs the same front door with the auto_round_amd plugin "
                               "(HIP engine), per case", "device": torch.cuda.get_device_name(0), "cases": recs}, f, indent=1)


if __name__ == "__main__":
    main()
