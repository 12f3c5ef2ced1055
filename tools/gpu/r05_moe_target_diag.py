"""Why do the reference-free flow's Mixtral targets differ from the reference's (tests/golden/t3s_mixtral8x7b_*.npz `y_sha`)?
Builder-side diagnostic (needs the staged reference tree): in ONE process, the same seeded Mixtral-8x7B-dimension block prepared
  (A) by the reference (`prepare_model_for_moe_quantization`: its linear_loop experts), and
  (B) by this package (`moe_unfuse.unfuse_moe_experts`),
forwarded over the same captured block inputs in minibatches of 8 the way both flows produce targets; compared with each other, run to
run, with torch's deterministic-algorithms mode on and off, and against the fixture's `y_sha`.  Drills into the first differing
sub-result (attention half, router, per-expert GEMM outputs).

    python tools/gpu/r05_moe_target_diag.py --out gpurun_out/r05/moe_target_diag.json
"""
import argparse
import copy
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def ndiff(a, b):
    if a.shape != b.shape or a.dtype != b.dtype:
        return -1
    it = {2: torch.int16, 4: torch.int32}[a.element_size()]
    return int((a.contiguous().view(it) != b.contiguous().view(it)).sum())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--arch", default="mixtral8x7b")
    ap.add_argument("--nsamples", type=int, default=64)
    ap.add_argument("--device", default="cuda:0")
    args = ap.parse_args()
    from ref_tree import import_reference

    import_reference()
    import auto_round.modeling.fused_moe.moe_experts_interface as mi

    from auto_round_amd.moe_unfuse import unfuse_moe_experts
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.testing import t3_fixture as fx

    dev = torch.device(args.device)
    res = dict(arch=args.arch, device=torch.cuda.get_device_name(0) if dev.type == "cuda" else "cpu")
    fixture = os.path.join(ROOT, "tests", "golden", "t3s_mixtral8x7b_mxfp4_100.npz")
    meta = json.loads(str(np.load(fixture, allow_pickle=False)["meta"])) if os.path.exists(fixture) and args.arch == "mixtral8x7b" else {}
    base = fx.build_model(args.arch)
    tokens = fx.calib_tokens(args.arch, args.nsamples, 2048 if args.arch == "mixtral8x7b" else 64)

    def prepare(kind):
        m = copy.deepcopy(base)
        if kind == "ref":
            mi.prepare_model_for_moe_quantization(m)
        m = m.to(dev)
        for p in m.parameters():
            p.requires_grad_(False)
        if kind == "ours":
            unfuse_moe_experts(m)
        blk = fx.decoder_blocks(m)[0]
        x0, others = fx.capture_block_inputs(m, blk, tokens, dev)
        return m, blk, x0, others

    if dev.type == "cuda":
        q = SignRoundQuantizer(SignRoundConfig(iters=1, batch_size=8, bits=4, sdpa_backend="auto", fused_block=False, exact_rounding=False), device=dev)
    else:           # (CPU dry run of this script only: the free function the quantizer's module path wraps)
        import types

        from auto_round_amd.quantizer import block_forward as _bf

        q = types.SimpleNamespace(block_forward=lambda b, x, o: _bf(b, x, o, amp=True, amp_dtype=torch.bfloat16),
                                  forward_all=lambda b, x, o: torch.cat([_bf(b, x[i:i + 8], o, amp=True, amp_dtype=torch.bfloat16) for i in range(0, x.shape[0], 8)]))

    def targets(blk, x0, others):
        if dev.type != "cuda":
            with torch.no_grad():
                return q.forward_all(blk, x0, others)
        with torch.cuda.device(dev):
            return q.forward_all(blk, x0, others)

    out = {}
    for det in (True, False):
        torch.use_deterministic_algorithms(det, warn_only=True)
        tag = "det_on" if det else "det_off"
        mA, bA, xA, oA = prepare("ref")
        yA1 = targets(bA, xA, oA)
        yA2 = targets(bA, xA, oA)
        shaA = fx.sha(yA1)
        del mA
        mB, bB, xB, oB = prepare("ours")
        yB1 = targets(bB, xB, oB)
        yB2 = targets(bB, xB, oB)
        rec = dict(x_identical=fx.sha(xA) == fx.sha(xB), x_matches_fixture=(fx.sha(xA) == meta.get("x_sha")) if meta else None,
                   ref_prep_run_to_run_differing=ndiff(yA1, yA2), ours_run_to_run_differing=ndiff(yB1, yB2),
                   ref_prep_vs_ours_differing=ndiff(yA1, yB1), numel=yA1.numel(), dtype=str(yA1.dtype),
                   ref_prep_matches_fixture_y=(shaA == meta.get("y_sha")) if meta else None,
                   ours_matches_fixture_y=(fx.sha(yB1) == meta.get("y_sha")) if meta else None)
        if rec["ref_prep_vs_ours_differing"]:
            d = (yA1.float() - yB1.float()).abs()
            rec["max_abs_diff"] = float(d.max())
            per_sample = [int((yA1[i].view(torch.int16) != yB1[i].view(torch.int16)).sum()) for i in range(yA1.shape[0])]
            rec["differing_per_sample_first16"] = per_sample[:16]
            # drill: first minibatch, hooks on both blocks
            grabs = {}

            def hook(name, store):
                def f(mod, inp, outp):
                    o = outp[0] if isinstance(outp, (tuple, list)) else outp
                    if isinstance(o, torch.Tensor):
                        store[name] = o.detach().clone()
                return f

            for tagb, blk, x, o in (("A", bA, xA, oA), ("B", bB, xB, oB)):
                store = {}
                hs = []
                for n, mod in blk.named_modules():
                    if n in ("self_attn", "input_layernorm", "post_attention_layernorm", "mlp", "mlp.gate", "mlp.experts") or n.startswith("mlp.experts.") and n.count(".") == 3:
                        hs.append(mod.register_forward_hook(hook(n, store)))
                with torch.no_grad():
                    q.block_forward(blk, x[:8], o)
                for h in hs:
                    h.remove()
                grabs[tagb] = store
            first = []
            for n in grabs["A"]:
                if n in grabs["B"]:
                    dd = ndiff(grabs["A"][n], grabs["B"][n])
                    if dd:
                        first.append((n, dd, list(grabs["A"][n].shape)))
            rec["differing_submodule_outputs_first_minibatch"] = first[:20]
            rec["submodules_compared"] = len(grabs["A"])
        out[tag] = rec
        print(tag, json.dumps(rec), flush=True)
        del mB
        if dev.type == "cuda":
            torch.cuda.empty_cache()
    res["modes"] = out
    if args.out:
        with open(args.out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
