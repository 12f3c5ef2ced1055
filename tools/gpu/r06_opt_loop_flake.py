"""Round 6 probe: one tuning-time forward + backward of the WRAPPED OPT-125M block (module path: transformers' code around this
package's quant kernels, library GEMMs and attention) replayed R times on the same minibatch and the same output gradient -- which
result, if any, is not reproducible?  Checksums (device side, no sync inside the loop) of the prediction, of every stage's output
and of every layer's weight gradient.  Then T full 200-iteration tunes with the reproducible no-grad attention forward on and off.

    python tools/gpu/r06_opt_loop_flake.py [R] [T]  ->  gpurun_out/r06/opt_loop_flake.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer  # noqa: E402
from auto_round_amd.schemes import apply_scheme, resolve_scheme  # noqa: E402
from auto_round_amd.testing import t3_fixture as fx  # noqa: E402
from auto_round_amd.wrapper import unwrapper_block, wrapper_block  # noqa: E402


def main():
    R = int(sys.argv[1]) if len(sys.argv) > 1 else 400
    T = int(sys.argv[2]) if len(sys.argv) > 2 else 10
    dev = torch.device("cuda:0")
    out = {"torch": torch.__version__, "device": torch.cuda.get_device_name(0), "R": R, "T": T}
    torch.use_deterministic_algorithms(True, warn_only=True)
    torch.utils.deterministic.fill_uninitialized_memory = False        # as inside quantize_block
    model = fx.build_model("opt125m").to(dev)
    for p in model.parameters():
        p.requires_grad_(False)
    tokens = fx.calib_tokens("opt125m", 16, 2048)
    block = fx.decoder_blocks(model)[0]
    apply_scheme(block, resolve_scheme("W4A16"))
    x0, others = fx.capture_block_inputs(model, block, tokens, dev)
    q = SignRoundQuantizer(SignRoundConfig(iters=200, batch_size=8, bits=4, sdpa_backend="auto"), device=dev)
    wrapper_block(block, True, False, enable_torch_compile=False, device=dev, iters=200)
    arenas = block._ar_arenas
    for a in arenas:
        a.qdq_forward()
    x = x0[:8].clone()
    g = torch.Generator(device=dev).manual_seed(7)
    dpred = (torch.randn(x.shape, generator=g, device=dev, dtype=torch.float32) * 1e-3).to(torch.bfloat16)
    names = list(fx.OPT_STAGES[:-1])
    layer_names = [n for n, _ in block.named_modules() if n.endswith(("q_proj", "k_proj", "v_proj", "out_proj", "fc1", "fc2"))]
    cols = ["pred"] + ["fwd:" + s for s in names] + ["dW:arena%d" % i for i in range(len(arenas))]
    sums = torch.zeros(R, len(cols), dtype=torch.int64, device=dev)
    mods = dict(block.named_modules())
    cur = {"i": 0}
    hs = []

    def out_hook(ci):
        def f(mod, inp, o):
            sums[cur["i"], ci].copy_(fx.bits_checksum(o[0] if isinstance(o, tuple) else o))
        return f

    for s in names:
        if s in mods:
            hs.append(mods[s].register_forward_hook(out_hook(cols.index("fwd:" + s))))
    def core_hook(mod, args):
        sums[cur["i"], cols.index("fwd:attn_core")].copy_(fx.bits_checksum(args[0]))

    hs.append(mods["self_attn.out_proj"].register_forward_pre_hook(core_hook))
    for i in range(R):
        cur["i"] = i
        for a in arenas:
            for l in a.layers:
                l._dw_accum[0] = False
        pred = q.block_forward(block, x, others)
        sums[i, 0].copy_(fx.bits_checksum(pred))
        pred.backward(dpred)
        for ai, a in enumerate(arenas):
            sums[i, cols.index("dW:arena%d" % ai)].copy_(fx.bits_checksum(a.dWq))
    torch.cuda.synchronize()
    for h in hs:
        h.remove()
    s = sums.cpu().numpy()
    rep = {}
    for ci, c in enumerate(cols):
        vals, cnt = np.unique(s[:, ci], return_counts=True)
        rep[c] = {"off_majority": int(R - cnt.max()), "distinct": int(len(vals))}
    out["replay"] = rep
    print("replay", {c: v["off_majority"] for c, v in rep.items()}, flush=True)
    unwrapper_block(block, {})
    del model, block, x0
    torch.cuda.empty_cache()

    fixp = os.path.join(ROOT, "tests", "golden", "t3s_opt125m_w4g128.npz")
    for repro in (True, False):
        runs = []
        for t in range(T):
            r = fx.check_against_stat_fixture(fixp, reproducible_attention=repro)
            runs.append({k: r[k] for k in ("bit_identical", "targets_identical", "first_divergence_iter", "first_differing_stage", "prefix_identical_codes",
                                           "best_loss_ratio", "tune_s")})
            print("tune reproducible_attention=%s" % repro, t, runs[-1], flush=True)
        out["tunes_reproducible_attention_%s" % repro] = {"runs": runs, "bit_identical": sum(r["bit_identical"] for r in runs),
                                                          "targets_identical": sum(r["targets_identical"] for r in runs), "of": T}
    os.makedirs(os.path.join(ROOT, "gpurun_out", "r06"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06", "opt_loop_flake.json"), "w") as f:
        json.dump(out, f, indent=1, default=str)


if __name__ == "__main__":
    main()
