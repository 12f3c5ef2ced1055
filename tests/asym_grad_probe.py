"""(test infrastructure, GPU) Where do the asymmetric INT scale gradients of `ar_qdq_int_bwd` leave the bits torch autograd
produces on the GPU?  Autograd of the pinned restatement (oracle/torch_ref.qdq_int == the reference's quant_tensor_asym) vs the
kernel vs a step-by-step manual evaluation of the same chain with torch.sum (torch's own reduction order) and with explicit pairwise
trees.  usage: python tests/asym_grad_probe.py [bits gs]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from auto_round_amd import ops
from oracle import torch_ref as tr


def tree_sum(t):
    """pairwise tree over the last dim (neighbours first)"""
    while t.shape[-1] > 1:
        t = t[..., 0::2] + t[..., 1::2]
    return t[..., 0]


def beq(a, b):
    return float((a.float().contiguous().view(torch.int32) == b.float().contiguous().view(torch.int32)).float().mean())


def main():
    bits = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    gs = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    out_f, in_f = 4096, 4096
    dev = "cuda"
    g0 = torch.Generator(device=dev).manual_seed(0)
    W = (torch.randn(out_f, in_f, generator=g0, device=dev) * 0.02).to(torch.bfloat16)
    G = out_f * in_f // gs
    V = (torch.rand(G, gs, generator=g0, device=dev) - 0.5).requires_grad_(True)
    mn = (0.9 + 0.1 * torch.rand(G, generator=g0, device=dev)).requires_grad_(True)
    mx = (0.9 + 0.1 * torch.rand(G, generator=g0, device=dev)).requires_grad_(True)
    dWq = (torch.randn(out_f, in_f, generator=g0, device=dev) * 1e-3).to(torch.bfloat16)
    Wg = W.reshape(-1, gs)
    wmin = torch.clamp(Wg.min(1)[0], max=0)
    wmax = torch.clamp(Wg.max(1)[0], min=0)
    Wq, s, zp = tr.qdq_int(W, bits, gs, False, V, mn, mx, wmin, wmax)
    Wq.backward(dWq)
    dV_k, dmin_k, dmax_k = ops.qdq_int_bwd(dWq.view(-1), W.view(-1), V.detach().reshape(-1).contiguous(), wmin, wmax, mn.detach(), mx.detach(),
                                           gs=gs, bits=bits, sym=0)
    rec = {"bits": bits, "gs": gs, "kernel_vs_autograd": {"dV": beq(dV_k, V.grad.reshape(-1)), "dmin": beq(dmin_k, mn.grad), "dmax": beq(dmax_k, mx.grad)}}
    # ---- manual chain
    with torch.no_grad():
        maxq = 2 ** bits - 1
        lo, hi = wmin * mn, wmax * mx                         # fp32 [G]
        s_raw = ((hi - lo) / maxq).to(torch.float16)
        s0 = torch.clamp(s_raw, min=1e-5)
        zp0 = torch.round(-lo / s0)
        sU, zU = s0.unsqueeze(-1), zp0.unsqueeze(-1)
        g = dWq.reshape(-1, gs).float()
        x = Wg / sU                                           # bf16 / fp16 -> fp32
        r = (torch.round(x + V) - (x + V)) + (x + V)
        t = r + zU
        inside = (t >= 0) & (t <= maxq)
        qq = torch.clamp(t, 0, maxq) - zU
        e = g * sU
        dy = torch.where(inside, e, torch.zeros_like(e))
        rec["manual_dV_vs_autograd"] = beq(dy, V.grad)
        bad = (dV_k.view(torch.int32) != V.grad.reshape(-1).view(torch.int32)).nonzero().reshape(-1)[:6]
        rec["dV_examples"] = []
        for i in bad.tolist():
            gi, k = divmod(i, gs)
            rec["dV_examples"].append(dict(i=i, W=float(Wg[gi, k]), V=float(V[gi, k]), s=float(s0[gi]), zp=float(zp0[gi]), x=float(x[gi, k]),
                                           y=float(x[gi, k] + V[gi, k]), r=float(r[gi, k]), t=float(t[gi, k]), g=float(g[gi, k]),
                                           dV_autograd=float(V.grad[gi, k]), dV_kernel=float(dV_k[i]), lo=float(lo[gi]), hi=float(hi[gi]),
                                           s_raw=float(s_raw[gi]), mn=float(mn[gi]), mx=float(mx[gi]), wmin=float(wmin[gi]), wmax=float(wmax[gi])))
        for tag, S in (("torchsum", lambda a: a.sum(-1)), ("tree", tree_sum)):
            c1 = S(g * qq).to(torch.float16)
            c2 = S((-dy) * (x / sU)).to(torch.float16)
            dzp = S(-e) + S(dy)
            c3 = ((-dzp) * (((-lo) / s0) / s0)).to(torch.float16)
            ds = ((c1 + c2) + c3)
            ds = torch.where(s_raw >= 1e-5, ds, torch.zeros_like(ds)).float()
            d32 = ds / maxq
            dlo = (-d32) + (-(dzp / s0))
            dmin_m, dmax_m = dlo * wmin, d32 * wmax
            rec[f"manual_{tag}_vs_autograd"] = {"dmin": beq(dmin_m, mn.grad), "dmax": beq(dmax_m, mx.grad)}
            if tag == "torchsum":
                badg = (dmax_k.view(torch.int32) != mx.grad.view(torch.int32)).nonzero().reshape(-1)[:5]
                rec["dmax_examples"] = [dict(g=int(j), c1=float(c1[j]), c2=float(c2[j]), c3=float(c3[j]), dzp=float(dzp[j]), ds=float(ds[j]), d32=float(d32[j]),
                                             dmax_autograd=float(mx.grad[j]), dmax_kernel=float(dmax_k[j]), dmin_autograd=float(mn.grad[j]),
                                             dmin_kernel=float(dmin_k[j]), s=float(s0[j]), zp=float(zp0[j]), wmax=float(wmax[j]), wmin=float(wmin[j]))
                                        for j in badg.tolist()]
            rec[f"manual_{tag}_vs_kernel"] = {"dmin": beq(dmin_m, dmin_k), "dmax": beq(dmax_m, dmax_k)}
            rec[f"pieces_{tag}"] = {"dzp_a_eq_tree": beq(S(-e), tree_sum(-e)), "dzp_b_eq_tree": beq(S(dy), tree_sum(dy)),
                                    "c1_eq_tree": beq(S(g * qq), tree_sum(g * qq)), "c2_eq_tree": beq(S((-dy) * (x / sU)), tree_sum((-dy) * (x / sU)))}
    print(json.dumps(rec))


if __name__ == "__main__":
    main()
