#!/usr/bin/env python
"""bench.py -- blocks tuned per second on MI355X (BASELINE.json metric), one process per GPU.

    python bench.py [--gpus N --steps K --warmup W]            (N>1: launched by torch.distributed.run, RCCL)

A "step" is ONE transformer block tuned end to end through the hot path -- reference fp forward of the 128
calibration samples, 200 sign-SGD iterations (fake-quant forward, block forward/backward against the cached
activations, MSE loss, fused backward + sign-SGD), unwrap with the best parameters, quantised-output forward, and the
final int4 packing -- i.e. what the reference times per block in `_quantize_blocks` (compressors/orchestrator.py:
176-388, 792-794) plus packing.  Workload at N=1: BASELINE.json configs[1] = Llama-3-8B W4 group_size=128 sym,
iters=200, nsamples=128, seqlen=2048, batch 8 (random-init weights of that architecture, synthetic N(0,1) hidden
states: there is no network for checkpoints or datasets).

Multi-GPU (weak scaling): every rank tunes its own K blocks (independent blocks shard embarrassingly); the shared
calibration activations are broadcast from rank 0 over RCCL/xGMI inside the timed region; no other collective exists
on the data path.  value = (N*K blocks) / max-over-ranks time.

Extra objects on the JSON line:
  roofline      quant-forward kernel (k_int_fwd): algorithmic bytes (8 B/elem + 12 B/group) / average
                launch duration measured live with HIP events on the launch stream inside the timed region
  roofline_bwd_sgd   same for the fused backward + sign-SGD kernel (12 B/elem + 8 B/group, +4 B/elem on snapshot iters)
  cpu_baseline  oracle/torch_ref (torch restatement of the reference loop, kind "port") timed on the host cores on a
                bounded sample, rank 0 at N=1 only
"""
import argparse
import json
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

WORKLOADS = {
    # name: (hidden, ffn, heads, kv_heads, family, description)
    "llama3-8b": dict(hidden=4096, ffn=14336, heads=32, kv=8, family="llama",
                      desc="Llama-3-8B decoder block, W4 group_size=128 sym (BASELINE.json configs[1])"),
    "llama3-70b": dict(hidden=8192, ffn=28672, heads=64, kv=8, family="llama",
                       desc="Llama-3-70B decoder block, W4 group_size=128 sym (configs[3], one block per step)"),
    "opt-125m": dict(hidden=768, ffn=3072, heads=12, kv=12, family="opt",
                     desc="OPT-125M decoder block, W4 group_size=128 sym (BASELINE.json configs[0])"),
    "mixtral-8x7b": dict(hidden=4096, ffn=14336, heads=32, kv=8, family="moe", experts=8, top_k=2,
                         desc="Mixtral-8x7B-shaped sparse-MoE decoder block (8 experts, top-2, experts as nn.Linear "
                              "w1/w2/w3 as after the reference's fused-MoE unfusing; BASELINE.json configs[4])"),
    "mixtral-8x7b-hf": dict(hidden=4096, ffn=14336, heads=32, kv=8, family="moe_hf", experts=8, top_k=2,
                            desc="Mixtral-8x7B decoder block as transformers builds it (MixtralDecoderLayer; fused 3-D expert "
                                 "parameters unfused by auto_round_amd.moe_unfuse; BASELINE.json configs[4])"),
}
SCHEMES = ("W4A16", "W2A16G32", "MXFP4", "NVFP4", "MXFP4_W", "NVFP4_W")
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy ceiling


def build_block(w, bits, gs, sym, device, seed, attn="sdpa", scheme=None):
    torch.manual_seed(seed)
    if w["family"] == "moe":
        from auto_round_amd.testing.moe import build_moe_decoder_layer, set_scheme

        layer, rope, cfg = build_moe_decoder_layer(w["hidden"], w["ffn"], w["heads"], w["kv"], w["experts"], w["top_k"],
                                                   device=device, attn=attn, seed=seed)
        n_w = set_scheme(layer, scheme or "W4A16")
        if scheme is None:
            for m in layer.modules():
                if isinstance(m, torch.nn.Linear) and getattr(m, "bits", 16) < 16:
                    m.bits, m.group_size, m.sym = bits, gs, sym
        return layer, rope, cfg, n_w
    if w["family"] == "moe_hf":
        from transformers import MixtralConfig
        from transformers.models.mixtral.modeling_mixtral import MixtralDecoderLayer, MixtralRotaryEmbedding

        from auto_round_amd.moe_unfuse import unfuse_moe_experts
        from auto_round_amd.testing.moe import set_scheme

        cfg = MixtralConfig(hidden_size=w["hidden"], intermediate_size=w["ffn"], num_attention_heads=w["heads"],
                            num_key_value_heads=w["kv"], num_hidden_layers=1, vocab_size=32000, rope_theta=1e6,
                            max_position_embeddings=32768, num_local_experts=w["experts"], num_experts_per_tok=w["top_k"])
        cfg._attn_implementation = attn
        with torch.device(device):
            layer = MixtralDecoderLayer(cfg, 0).to(torch.bfloat16)
            rope = MixtralRotaryEmbedding(cfg)
        for p in layer.parameters():                        # torch.empty expert parameters: give them a weight-like scale
            if p.dim() == 3:
                p.data.normal_(0.0, 0.02)
        layer.eval()
        for p in layer.parameters():
            p.requires_grad_(False)
        unfuse_moe_experts(layer)
        n_w = set_scheme(layer, scheme or "W4A16")
        return layer, rope, cfg, n_w
    if w["family"] == "llama":
        from transformers import LlamaConfig
        from transformers.models.llama.modeling_llama import LlamaDecoderLayer, LlamaRotaryEmbedding

        cfg = LlamaConfig(hidden_size=w["hidden"], intermediate_size=w["ffn"], num_attention_heads=w["heads"],
                          num_key_value_heads=w["kv"], num_hidden_layers=1, vocab_size=128256, rope_theta=500000.0,
                          max_position_embeddings=8192)
        cfg._attn_implementation = attn
        with torch.device(device):
            layer = LlamaDecoderLayer(cfg, 0).to(torch.bfloat16)
            rope = LlamaRotaryEmbedding(cfg)
    else:
        from transformers import OPTConfig
        from transformers.models.opt.modeling_opt import OPTDecoderLayer

        cfg = OPTConfig(hidden_size=w["hidden"], ffn_dim=w["ffn"], num_attention_heads=w["heads"], num_hidden_layers=1,
                        vocab_size=50272, max_position_embeddings=2048, word_embed_proj_dim=w["hidden"])
        cfg._attn_implementation = attn
        with torch.device(device):
            layer = OPTDecoderLayer(cfg).to(torch.bfloat16)
        rope = None
    layer.eval()
    for p in layer.parameters():
        p.requires_grad_(False)
    n_w = 0
    for m in layer.modules():
        if isinstance(m, torch.nn.Linear):
            m.bits, m.group_size, m.sym, m.data_type, m.scale_dtype, m.act_bits = bits, gs, sym, "int", torch.float16, 16
            n_w += m.weight.numel()
    if scheme is not None:
        from auto_round_amd.testing.moe import set_scheme

        n_w = set_scheme(layer, scheme)
    return layer, rope, cfg, n_w


def make_others(rope, seqlen, device, x1):
    if rope is None:
        return {}
    pos = torch.arange(seqlen, device=device).unsqueeze(0)
    cos, sin = rope(x1, pos)
    return {"position_embeddings": (cos, sin), "attention_mask": None, "position_ids": pos}


class KernelTimer:
    """HIP-event timing of individual kernel launches on the launch stream (torch's current stream is the stream the
    C ABI receives), accumulated only while `enabled` (the timed region)."""

    def __init__(self):
        self.enabled = False
        self.pairs = {}

    def wrap(self, name, fn):
        def inner(*a, **k):
            if not self.enabled:
                return fn(*a, **k)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn(*a, **k)
            e.record()
            self.pairs.setdefault(name, []).append((s, e))
            return out
        return inner

    def mean_ms(self, name):
        ps = self.pairs.get(name, [])
        if not ps:
            return None, 0
        return sum(s.elapsed_time(e) for s, e in ps) / len(ps), len(ps)


def cpu_baseline(w, bits, gs, sym, args):
    """oracle/torch_ref (the pinned torch restatement of the reference loop) on the host cores, bounded sample:
    one tuning iteration at batch 1 and one at batch 2 of the real shapes -> linear extrapolation to the real batch
    (the fake-quant part does not depend on the batch; the GEMM/attention part is linear in tokens)."""
    from oracle import torch_ref as tr

    torch.manual_seed(0)
    layer, rope, cfg, n_w = build_block(w, bits, gs, sym, "cpu", seed=0)
    S, H = args.seqlen, w["hidden"]
    X = torch.randn(2, S, H).to(torch.bfloat16)
    others = make_others(rope, S, "cpu", X[:1])

    def fwd(blk, x, o):
        out = blk(x, **o)
        return out[0] if isinstance(out, (tuple, list)) else out

    tr.wrap_block(layer, True)
    wrappers = [m for m in layer.modules() if isinstance(m, tr.RefWrapperLinear)]
    params = [p for wr in wrappers for p in wr.params.values()]
    mse = torch.nn.MSELoss()

    def one_iter(b):
        x = X[:b]
        with torch.autocast("cpu", dtype=torch.bfloat16):
            out = fwd(layer, x, others)
        loss = mse(out.float(), x.float())
        (loss * 1000).backward()
        tr.sign_sgd_step(params, 0.005)
        for p in params:
            p.grad = None
        return loss.item()

    t0 = time.time(); one_iter(1); t_warm = time.time() - t0
    t0 = time.time(); one_iter(1); t1 = time.time() - t0
    t0 = time.time(); one_iter(2); t2 = time.time() - t0
    slope = max(t2 - t1, 0.0)
    t_iter = t1 + slope * (args.batch_size - 1)
    blocks_per_s = 1.0 / (args.iters * t_iter)
    return {"value": blocks_per_s, "unit": "blocks/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"oracle/torch_ref.py (torch restatement of the reference loop) on CPU, same block shapes: 1 warm-up "
                      f"+ 1 timed iteration at batch 1x{S} ({t1:.2f}s) and 1 at batch 2x{S} ({t2:.2f}s); per-iteration time "
                      f"extrapolated linearly to batch {args.batch_size} ({t_iter:.2f}s) x {args.iters} iters; fp/q-output "
                      f"forwards and packing not included (favours the CPU)",
            "sec_per_iter_at_batch": t_iter, "warmup_iter_s": t_warm}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="llama3-8b", choices=sorted(WORKLOADS))
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--nsamples", type=int, default=128)
    ap.add_argument("--seqlen", type=int, default=2048)
    ap.add_argument("--batch-size", type=int, default=8)
    ap.add_argument("--bits", type=int, default=4)
    ap.add_argument("--group-size", type=int, default=128)
    ap.add_argument("--asym", action="store_true")
    ap.add_argument("--scheme", default=None, choices=SCHEMES,
                    help="reference scheme preset (overrides --bits/--group-size/--asym); MXFP4/NVFP4 include 4-bit activations")
    ap.add_argument("--fuse-next-forward", action="store_true",
                    help="emit the next iteration's Wq from the fused backward kernel (K1 then runs once per block)")
    ap.add_argument("--sdpa", default="efficient", choices=["auto", "efficient", "flash", "math"],
                    help="SDPA backend priority for the block attention (see SignRoundConfig.sdpa_backend)")
    ap.add_argument("--data-parallel", action="store_true",
                    help="N>1: all ranks tune the SAME block, each on 1/N of every minibatch, all-reducing the weight-gradient "
                         "buffer per iteration (strong scaling; for quantised-input chaining). Default N>1 mode: block sharding")
    ap.add_argument("--alg-ext", action="store_true",
                    help="tune with the algorithm extension (SignRoundV2: imatrix, searched init scales, outlier loss)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # AR_BENCH_ONE_DEVICE_DEBUG=1: run every rank on cuda:0 over gloo -- only to exercise the N>1 code path on a 1-GPU box
    one_dev_debug = os.environ.get("AR_BENCH_ONE_DEVICE_DEBUG") == "1"
    if one_dev_debug:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if one_dev_debug:
            dist_mod.init_process_group("gloo")
        else:
            dist_mod.init_process_group("nccl", device_id=device)   # "nccl" is RCCL on ROCm
        dist = dist_mod

    from auto_round_amd import _lib, ops

    if not os.path.exists(_lib.LIB_PATH):      # fresh checkout: compile the HIP library first (no other code path exists)
        if rank == 0:
            print(f"[bench] {_lib.LIB_PATH} missing -> building with hipcc", file=sys.stderr)
            _lib.build()
        if dist is not None:
            dist.barrier()
    from auto_round_amd.export import pack_block
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer, SignRoundV2Quantizer

    w = WORKLOADS[args.workload]
    sym = not args.asym
    attn = "sdpa"
    if args.sdpa == "efficient":     # explicit K/V head repeat so that the efficient SDPA kernels are eligible for GQA
        from auto_round_amd.attention import register_mi355x_sdpa

        attn = register_mi355x_sdpa()
    dp = bool(args.data_parallel and world > 1)
    seed_rank = 0 if dp else rank                           # data parallel: every rank holds the same block and samples
    layer, rope, cfg, n_w = build_block(w, args.bits, args.group_size, sym, device, seed=1234 + seed_rank, attn=attn,
                                        scheme=args.scheme)
    fp4 = args.scheme is not None and args.scheme.startswith(("MXFP4", "NVFP4"))
    if args.scheme is not None:
        some = next(m for m in layer.modules() if isinstance(m, torch.nn.Linear) and getattr(m, "bits", 16) < 16)
        args.bits, args.group_size, sym = int(some.bits), int(some.group_size), bool(some.sym)
    master = {n: p.detach().clone() for n, p in layer.named_parameters()}
    S, H, N = args.seqlen, w["hidden"], args.nsamples
    X = torch.empty(N, S, H, dtype=torch.bfloat16, device=device)
    if rank == 0:
        g = torch.Generator(device=device).manual_seed(2)
        X.copy_(torch.randn(N, S, H, generator=g, device=device, dtype=torch.float32).to(torch.bfloat16))
    others = make_others(rope, S, device, X[:1])
    # token ids as the reference's calibrator caches them: the last position of every sample is marked -100 and is
    # excluded from the loss (calibration/llm.py:340-360) -> the masked loss path is the reference's default path
    token_ids = torch.randint(0, 32000, (N, S), generator=torch.Generator().manual_seed(3))
    token_ids[:, -1] = -100

    timer = KernelTimer()
    if not args.no_kernel_timing:
        ops.qdq_int_fwd = timer.wrap("k_int_fwd", ops.qdq_int_fwd)
        ops.qdq_int_bwd_sgd_ = timer.wrap("k_int_bwd_sgd", ops.qdq_int_bwd_sgd_)
        ops.qdq_fp4_bwd_sgd_ = timer.wrap("k_fp4_bwd_sgd", ops.qdq_fp4_bwd_sgd_)
        _fp4_fwd = ops.qdq_fp4_fwd
        _timed_w = timer.wrap("k_fp4_fwd", _fp4_fwd)
        # weight launches pass absmax; activation fake-quant launches (absmax=None) are not the roofline kernel
        ops.qdq_fp4_fwd = lambda X_, V_, absmax_, *a_, **k_: (_timed_w if absmax_ is not None else _fp4_fwd)(X_, V_, absmax_, *a_, **k_)

    qcfg = SignRoundConfig(iters=args.iters, batch_size=args.batch_size, bits=args.bits,
                           fuse_next_forward=args.fuse_next_forward, sdpa_backend=args.sdpa, data_parallel=dp)
    quantizer = (SignRoundV2Quantizer if args.alg_ext else SignRoundQuantizer)(qcfg, device=device)
    random.seed(42 + seed_rank)

    def restore():
        from auto_round_amd.wrapper import WrapperWALayer, _set_module

        for n, m in list(layer.named_modules()):            # A4 schemes leave activation-quant shells around the layers
            if isinstance(m, WrapperWALayer):
                _set_module(layer, n, m.orig_layer)
        with torch.no_grad():
            for n, p in layer.named_parameters():
                if p.data.shape == master[n].shape:
                    p.data = master[n].clone()

    def one_block():
        restore()                                           # "dispatch_block": fresh fp weights in HBM
        fp_out, q_out, best = quantizer.compress_block(layer, X, others, input_ids=token_ids)
        packed = pack_block(layer)                          # final low-bit packing kernel, GPTQ-order int32 words
        return quantizer.last_stats, packed

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        one_block()
    barrier()
    timer.enabled = True
    t0 = time.perf_counter()
    if dist is not None:
        dist.broadcast(X, src=0)                            # shared calibration activations over RCCL/xGMI
    stats = None
    for _ in range(args.steps):
        stats, packed = one_block()
    barrier()
    elapsed = time.perf_counter() - t0
    timer.enabled = False
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        G = n_w // args.group_size
        default_cfg = (args.scheme is None and args.bits == 4 and args.group_size == 128 and sym and args.iters == 200
                       and N == 128 and S == 2048)
        # the description names the BASELINE config only when the run really is that config
        workload_desc = w["desc"] if default_cfg else (
            w["desc"].split(",")[0] + f", {args.scheme or ('W%dG%d %s' % (args.bits, args.group_size, 'sym' if sym else 'asym'))}"
            f", iters={args.iters}, calib {N}x{S} (non-default variant of the BASELINE config)")
        out = {
            "metric": "transformer blocks tuned/sec (200 iters, 128x2048 calib)",
            "value": (1 if dp else world) * args.steps / elapsed,
            "unit": "blocks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1000.0 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "strong" if dp else "weak", "vs_baseline": None,
            "dtype": "bf16",
            "dtype_detail": "bf16 weights/activations and MFMA GEMMs (fp32 accumulate); fp32 rounding offsets V and min/max "
                            "scales; fp16 quant scales; int4 packed output",
            "data": "synthetic: random-init weights of the named architecture, N(0,1) bf16 hidden states",
            "config": {"workload": workload_desc, "scheme": args.scheme or "int", "bits": args.bits, "group_size": args.group_size, "sym": sym,
                       "iters": args.iters, "nsamples": N, "seqlen": S, "batch_size": args.batch_size,
                       "weights_per_block": n_w, "groups_per_block": G, "includes_packing": True,
                       "fuse_next_forward": bool(args.fuse_next_forward), "sdpa_backend": args.sdpa,
                       "alg_ext": bool(args.alg_ext),
                       "parallelism": f"data-parallel inside the block x{world}" if dp else f"block-sharded x{world}"},
            "ms_per_iter": 1000.0 * elapsed / args.steps / max(args.iters, 1),
            "loss": {"init": stats["init_loss"], "best": stats["best_loss"], "best_iter": stats["best_iter"]},
        }
        if fp4:
            fname, bname, per_g_f, per_g_b = "k_fp4_fwd", "k_fp4_bwd_sgd", 8, 8
        else:
            fname, bname, per_g_f, per_g_b = "k_int_fwd", "k_int_bwd_sgd", 12, 8
        ms, cnt = timer.mean_ms(fname)
        # only block-wide launches count (the per-layer unwrap calls are smaller): filter by duration is fragile,
        # so the block-wide figure is taken from the launches made by the arena (count = iters per step)
        if ms is not None:
            fwd_ms = timer_block_mean(timer, fname, n_w)
            abytes = 8 * n_w + per_g_f * G
            if fp4 and args.scheme.startswith("NVFP4"):   # NVFP4 launches per layer (own global scale): use the byte-weighted mean
                tot = sum(s.elapsed_time(e) for s, e in timer.pairs[fname])
                launches_per_block = 4 + 3 * w.get("experts", 0) if w["family"] in ("moe", "moe_hf") else 7
                fwd_ms = tot / (len(timer.pairs[fname]) / launches_per_block)
            out["roofline"] = {"kernel": fname + (" (fp4 weight fake-quant forward)" if fp4 else
                                                   " (INT fake-quant forward, whole block per launch)"), "bound": "hbm",
                               "achieved": abytes / fwd_ms / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": abytes / fwd_ms / 1e6 / HBM_PEAK_GBPS, "traffic": read_traffic(fname, abytes),
                               "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": fwd_ms}
        ms2, cnt2 = timer.mean_ms(bname)
        if ms2 is not None:
            per = 12 + (2 if (args.fuse_next_forward and not fp4) else 0)
            abytes = per * n_w + per_g_b * G
            if fp4 and args.scheme.startswith("NVFP4"):
                launches_per_block = 4 + 3 * w.get("experts", 0) if w["family"] in ("moe", "moe_hf") else 7
                ms2 = ms2 * launches_per_block
            out["roofline_bwd_sgd"] = {"kernel": ("k_fp4_bwd" if fp4 else "k_int_bwd") + " (fused qdq backward + sign-SGD" +
                                       (" + next forward)" if (args.fuse_next_forward and not fp4) else ")"), "bound": "hbm",
                                       "achieved": abytes / ms2 / 1e6, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                       "frac": abytes / ms2 / 1e6 / HBM_PEAK_GBPS,
                                       "traffic": read_traffic("k_int_bwd_with_next_fwd" if args.fuse_next_forward else "k_int_bwd", abytes),
                                       "algorithmic_bytes_per_launch": abytes, "avg_launch_ms": ms2, "launches": cnt2}
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(w, args.bits, args.group_size, sym, args)
                out["speedup_vs_cpu_baseline"] = out["value"] / out["cpu_baseline"]["value"]
            except Exception as e:  # pragma: no cover
                out["cpu_baseline"] = {"error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


def timer_block_mean(timer, name, n_w):
    """Mean duration of the block-wide launches only: the 7 per-layer unwrap launches of each step are much shorter
    than a whole-block launch, so the block-wide ones are the upper cluster (>= half of the maximum)."""
    d = [s.elapsed_time(e) for s, e in timer.pairs[name]]
    top = max(d)
    big = [x for x in d if x >= 0.5 * top]
    return sum(big) / len(big)


def read_traffic(kernel, abytes=None):
    """HBM bytes per launch from the committed PMC passes (profiles/pmc_traffic.json), or null.  The PMC run was made at
    the Llama-3-8B g128 block size; it is only reported when this run launches the same number of algorithmic bytes."""
    p = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(p) as f:
            d = json.load(f)
        if abytes is not None and d.get("algorithmic", {}).get(kernel) != abytes:
            return None
        return d.get(kernel)
    except Exception:
        return None


if __name__ == "__main__":
    main()
