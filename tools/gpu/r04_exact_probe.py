"""Round-4 probe behind `exact_rounding`: which ops of the Llama block reproduce eager torch bit for bit.

  python tools/gpu/r04_exact_probe.py [--stage kernels,gemms,plan,time,digest] [--out gpurun_out/r04_exact_probe.json]

stages: kernels  csrc/ar_exact.hip against the eager op chains (transformers' own modules under autograd), mismatch counts
        gemms    library GEMM forms at the Llama-3-8B minibatch: merged vs separate, transposed-copy dX, MFMA dW (unsplit / split)
        plan     ExactLlamaBlock.plan_against_module at Llama-3-8B dimensions (what is kept)
        time     ms per forward+backward: module path / exact plan / fused path
        digest   the full 200-iteration recipe against the reference-made digest with exact_rounding
Every stage is wrapped: a failure is recorded and the next stage runs."""
import argparse
import json
import os
import sys
import time
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402

DEV = "cuda:0"
BF = torch.bfloat16


def bits(t):
    return t.contiguous().view({2: torch.int16, 4: torch.int32}[t.element_size()])


def diff(a, b):
    """mismatching elements + the largest difference in units of the last place (bf16 / fp32 bit patterns, same sign)"""
    ne = bits(a) != bits(b)
    n = int(ne.sum())
    if n == 0:
        return dict(equal=True, mismatch=0, of=a.numel())
    d = (bits(a).to(torch.int64) - bits(b).to(torch.int64)).abs()
    return dict(equal=False, mismatch=n, of=a.numel(), max_ulp=int(d[ne].max()), frac=n / a.numel())


def stage_kernels():
    from transformers.models.llama.modeling_llama import LlamaRMSNorm, apply_rotary_pos_emb

    out = {}
    g = torch.Generator(device=DEV).manual_seed(0)
    for (T, H) in ((64, 256), (128, 768), (2048, 3584), (16384, 4096), (4096, 8192), (512, 8320)):
        norm = LlamaRMSNorm(H, 1e-5).to(BF).to(DEV)
        with torch.no_grad():
            norm.weight.copy_(1 + 0.2 * torch.randn(H, device=DEV, generator=g))
        x = torch.randn(T, H, device=DEV, generator=g).to(BF)
        r = (0.5 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
        dy = (0.01 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
        dres = (0.01 * torch.randn(T, H, device=DEV, generator=g)).to(BF)
        s = x + r
        sl = s.detach().requires_grad_(True)
        y = norm(sl.view(1, T, H))
        y.backward(dy.view(1, T, H))
        rstd_t = torch.rsqrt(s.float().pow(2).mean(-1, keepdim=True) + 1e-5).view(-1)
        res = ops.rmsnorm_fwd_exact(x, norm.weight.detach(), 1e-5, res=r)
        rec = {}
        if res is None:
            rec["refused"] = True
        else:
            y2, rstd, s2 = res
            w = norm.weight.detach()
            raw = ops.rmsnorm_fwd_exact(s, w, 1e-5, raw_sum=True)[1]
            rec = dict(sum=diff(s2, s), row_sum=diff(raw, s.float().pow(2).sum(-1)), rstd=diff(rstd, rstd_t),
                       rstd_f32_instruction=diff(ops.rmsnorm_fwd_exact(s, w, 1e-5, rsqrt_f32=True)[1], rstd_t),
                       y=diff(y2, y.detach().view(T, H)),
                       dx_own_rstd=diff(ops.rmsnorm_bwd_exact(dy, s2, w, rstd), sl.grad),
                       dx_torch_rstd=diff(ops.rmsnorm_bwd_exact(dy, s, w, rstd_t.contiguous()), sl.grad),
                       dx_plus_res=diff(ops.rmsnorm_bwd_exact(dy, s, w, rstd_t.contiguous(), dres=dres), sl.grad + dres))
            rec["y_nores"] = diff(ops.rmsnorm_fwd_exact(s, w, 1e-5)[0], y.detach().view(T, H))
        out[f"rmsnorm_{T}x{H}"] = rec
    # rotary
    for (B, S, hq, hkv, d) in ((2, 128, 4, 2, 64), (8, 2048, 32, 8, 128)):
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
        q2, k2 = ops.rope_fwd_exact(q, k, cos, sin, S, hq, hkv, d)
        dq2, dk2 = ops.rope_bwd_exact(gq, gk, cos, sin, S, d)
        gq_t = gq.transpose(1, 2).contiguous().transpose(1, 2)      # token-major strides, as the flash backward leaves them
        dq3, _ = ops.rope_bwd_exact(gq_t, gk, cos, sin, S, d)
        out[f"rope_{B}x{S}x{hq}x{hkv}x{d}"] = dict(
            q=diff(q2, qr.detach().transpose(1, 2).reshape(T, -1)), k=diff(k2, kr.detach().transpose(1, 2).reshape(T, -1)),
            dq=diff(dq2, dq.transpose(1, 2).reshape(T, -1)), dk=diff(dk2, dk.transpose(1, 2).reshape(T, -1)),
            dq_strided=diff(dq3, dq.transpose(1, 2).reshape(T, -1)))
    # SwiGLU: every bf16 gate value x 64 up values; backward with both forms of silu_backward's inner expression
    allb = torch.arange(65536, device=DEV, dtype=torch.int32).to(torch.int16).view(BF)
    allb = allb[torch.isfinite(allb.float())]
    n = allb.numel() // 8 * 8
    gate = allb[:n].repeat(64, 1).contiguous()
    up = torch.randn(64, n, device=DEV, generator=g).to(BF)
    da = torch.randn(64, n, device=DEV, generator=g).to(BF)
    gl, ul = gate.clone().requires_grad_(True), up.clone().requires_grad_(True)
    a = torch.nn.functional.silu(gl) * ul
    dgl, dul = torch.autograd.grad(a, (gl, ul), da)
    a2 = ops.swiglu_fwd_exact(gate, up)
    rec = dict(fwd=diff(a2, a.detach()))
    for c in (True, False):
        dg2, du2 = ops.swiglu_bwd_exact(da, gate, up, contract=c)
        rec[f"dg_contract_{int(c)}"] = diff(dg2, dgl)
        rec[f"du_contract_{int(c)}"] = diff(du2, dul)
    # random realistic magnitudes too
    gate = torch.randn(4096, 14336, device=DEV, generator=g).to(BF)
    up = torch.randn(4096, 14336, device=DEV, generator=g).to(BF)
    da = (0.01 * torch.randn(4096, 14336, device=DEV, generator=g)).to(BF)
    gl, ul = gate.clone().requires_grad_(True), up.clone().requires_grad_(True)
    a = torch.nn.functional.silu(gl) * ul
    dgl, dul = torch.autograd.grad(a, (gl, ul), da)
    rec["fwd_random"] = diff(ops.swiglu_fwd_exact(gate, up), a.detach())
    for c in (True, False):
        dg2, du2 = ops.swiglu_bwd_exact(da, gate, up, contract=c)
        rec[f"dg_random_contract_{int(c)}"] = diff(dg2, dgl)
    out["swiglu"] = rec
    return out


def timed(fn, n=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def stage_gemms():
    """Llama-3-8B minibatch (T = 16384 tokens): which faster GEMM forms return the module path's bits, and what they cost"""
    out = {}
    g = torch.Generator(device=DEV).manual_seed(1)
    T, H, Fd, KV = 16384, 4096, 14336, 1024

    def rnd(*shape, s=1.0):
        return (s * torch.randn(*shape, device=DEV, generator=g)).to(BF)

    x = rnd(T, H)
    Wq, Wk, Wv = rnd(H, H, s=0.02), rnd(KV, H, s=0.02), rnd(KV, H, s=0.02)
    Wqkv = torch.cat([Wq, Wk, Wv]).contiguous()
    y = torch.nn.functional.linear(x, Wqkv)
    sep = [torch.nn.functional.linear(x, w) for w in (Wq, Wk, Wv)]
    out["fwd_merged_qkv"] = dict(q=diff(y[:, :H], sep[0]), k=diff(y[:, H:H + KV], sep[1]), v=diff(y[:, H + KV:], sep[2]),
                                 ms_merged=timed(lambda: torch.nn.functional.linear(x, Wqkv)),
                                 ms_separate=timed(lambda: [torch.nn.functional.linear(x, w) for w in (Wq, Wk, Wv)]))
    Wg, Wu = rnd(Fd, H, s=0.02), rnd(Fd, H, s=0.02)
    Wgu = torch.cat([Wg, Wu]).contiguous()
    y = torch.nn.functional.linear(x, Wgu)
    out["fwd_merged_gu"] = dict(g=diff(y[:, :Fd], torch.nn.functional.linear(x, Wg)), u=diff(y[:, Fd:], torch.nn.functional.linear(x, Wu)),
                                ms_merged=timed(lambda: torch.nn.functional.linear(x, Wgu)),
                                ms_separate=timed(lambda: [torch.nn.functional.linear(x, w) for w in (Wg, Wu)]))
    del y, sep, Wgu, Wqkv
    # dX = dY W: as stored vs through a transposed copy
    for name, (o, i) in dict(o=(H, H), g=(Fd, H), d=(H, Fd)).items():
        W = rnd(o, i, s=0.02)
        dY = rnd(T, o, s=0.01)
        Wt = W.t().contiguous()
        a, b = torch.mm(dY, W), torch.mm(dY, Wt.t())
        out[f"dx_tn_{name}"] = dict(**diff(b, a), ms_nn=timed(lambda: torch.mm(dY, W)), ms_tn=timed(lambda: torch.mm(dY, Wt.t())),
                                    ms_transpose=timed(lambda: ops.transpose16(W, out=Wt)))
        del W, dY, Wt, a, b
    # dW = dY^T X: library vs the MFMA kernel (whole K in one pass / split plans)
    for name, (o, i) in dict(q=(H, H), k=(KV, H), o=(H, H), g=(Fd, H), d=(H, Fd), qkv=(H + 2 * KV, H), gu=(2 * Fd, H)).items():
        dY = rnd(T, o, s=0.01)
        X = rnd(T, i)
        ref = torch.mm(dY.t(), X)
        rec = dict(ms_lib=timed(lambda: torch.mm(dY.t(), X)))
        for split in (False, True):
            mine = torch.empty_like(ref)
            ok = ops.gemm_dw(dY, X, mine, split=split)
            rec[f"mfma_split_{int(split)}"] = dict(**diff(mine, ref), ms=timed(lambda: ops.gemm_dw(dY, X, mine, split=split))) if ok else "refused"
        out[f"dw_{name}"] = rec
        del dY, X, ref, mine
    return out


def stage_dw_split():
    """the library's weight-gradient GEMM vs the MFMA kernel with n contiguous K slices, at the shapes the one-pass form misses"""
    out = {}
    g = torch.Generator(device=DEV).manual_seed(2)
    T, H, Fd = 16384, 4096, 14336
    for name, (o, i) in dict(g=(Fd, H), d=(H, Fd), o=(H, H), big70b=(28672, 8192)).items():
        dY = (0.01 * torch.randn(T, o, device=DEV, generator=g)).to(BF)
        X = torch.randn(T, i, device=DEV, generator=g).to(BF)
        ref = torch.mm(dY.t(), X)
        rec = dict(ms_lib=timed(lambda: torch.mm(dY.t(), X)))
        for n in (1, 2, 3, 4, 7, 8):
            mine = torch.empty_like(ref)
            ok = ops.gemm_dw(dY, X, mine, split=(False if n == 1 else n))
            rec[f"slices_{n}"] = dict(**diff(mine, ref), ms=timed(lambda: ops.gemm_dw(dY, X, mine, split=(False if n == 1 else n)))) if ok else "refused"
        out[f"dw_{name}"] = rec
        del dY, X, ref, mine
    return out


def build_llama8b(nsamples=16, seqlen=2048):
    from auto_round_amd.quantizer import SignRoundConfig, SignRoundQuantizer
    from auto_round_amd.schemes import apply_scheme, resolve_scheme
    from auto_round_amd.testing import t3_fixture as fx

    model = fx.build_model("llama8b").to(DEV)
    for p in model.parameters():
        p.requires_grad_(False)
    tokens = fx.calib_tokens("llama8b", nsamples, seqlen)
    block = fx.decoder_blocks(model)[0]
    sch = resolve_scheme("W4A16")
    apply_scheme(block, sch)
    x0, others = fx.capture_block_inputs(model, block, tokens, torch.device(DEV))
    q = SignRoundQuantizer(SignRoundConfig(iters=200, batch_size=8, bits=4, sdpa_backend="auto"), device=DEV)
    y = q.calibrate_block(block, x0, others)
    return block, x0, others, y, q


def stage_plan_and_time(with_mask=True):
    from auto_round_amd.exact_block import ExactLlamaBlock
    from auto_round_amd.fused_block import build_fused_block
    from auto_round_amd.wrapper import wrapper_block

    out = {}
    block, x0, others, y, q = build_llama8b()
    if not with_mask:
        others = {k: v for k, v in others.items() if k != "attention_mask"}
        y = q.calibrate_block(block, x0, others)
    wrapper_block(block, True, False, enable_torch_compile=False, device=torch.device(DEV), iters=200)
    arenas = block._ar_arenas
    eb = ExactLlamaBlock.try_build(block, arenas, others, BF, sdpa_ctx=q._sdpa_ctx, amp=True)
    out["recognised"] = eb is not None
    if eb is None:
        return out
    mf = lambda x, o: q.block_forward(block, x, o)  # noqa: E731
    t0 = time.perf_counter()
    plan = eb.plan_against_module(mf, x0[:8].clone(), others, y[:8])
    torch.cuda.synchronize()
    out["plan_seconds"] = time.perf_counter() - t0
    out["report"] = eb.plan_report
    # one iteration's forward + backward, three ways (K1 / K2 / loss excluded: identical in all three)
    x = x0[:8].clone()
    dpred = (1e-3 * torch.randn_like(x))

    def run_module():
        for a in arenas:
            for l in a.layers:
                l._dw_accum[0] = False
        p = mf(x, others)
        p.backward(dpred)

    out["ms_fwd_bwd_module"] = timed(run_module)
    if plan:
        out["ms_fwd_bwd_exact"] = timed(lambda: eb._run_once(x, others, dpred))
        allk = dict(plan)
        for k in ("norm1", "norm2", "rope", "swiglu"):
            allk[k] = False
        eb.set_plan(allk)
        out["ms_fwd_bwd_exact_torch_elementwise"] = timed(lambda: eb._run_once(x, others, dpred))
        eb.set_plan(plan)
    for a in arenas:
        for l in a.layers:
            l._mfma_dw = True
    fb = build_fused_block(block, arenas, others, BF, sdpa_ctx=q._sdpa_ctx, use_mfma_dw=True, tn_dx_gemm=True)
    if fb is not None:
        def run_fused():
            for a in arenas:
                for l in a.layers:
                    l._dw_accum[0] = False
            p = fb.forward(x.clone(), others)
            p.backward(dpred)
        out["ms_fwd_bwd_fused"] = timed(run_fused)
    return out


def stage_digest():
    from auto_round_amd.testing import t3_fixture as fx

    r = fx.check_against_digest(exact=True)
    r["blocks_per_s_incl_plan"] = 1.0 / r["tune_s"]
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--stage", default="kernels,plan,digest")
    ap.add_argument("--out", default="gpurun_out/r04_exact_probe.json")
    a = ap.parse_args()
    res = dict(device=torch.cuda.get_device_name(0), torch=torch.__version__)
    stages = dict(kernels=stage_kernels, gemms=stage_gemms, dw_split=stage_dw_split, plan=lambda: stage_plan_and_time(True),
                  plan_nomask=lambda: stage_plan_and_time(False), digest=stage_digest)
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    for name in a.stage.split(","):
        t0 = time.perf_counter()
        try:
            res[name] = stages[name]()
        except Exception:  # noqa: BLE001
            res[name] = dict(error=traceback.format_exc()[-3000:])
        res[name + "_seconds"] = time.perf_counter() - t0
        torch.cuda.empty_cache()
        with open(a.out, "w") as f:
            json.dump(res, f, indent=1, default=str)
        print(f"[{name}] {res[name + '_seconds']:.1f}s", flush=True)
    print(json.dumps(res, default=str)[:6000])


if __name__ == "__main__":
    main()
