"""A few launches of the MFMA kernels of this repository (and the library kernels next to them) for `rocprofv3 --pmc` passes
(MFMA utilisation / LDS stall counters for k_gemm_dw4, k_attn_fwd, k_attn_bwd):

    bash tools/gpu/run.sh pmc mfma "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE" -- tools/gpu/r04_pmc_kernels.py
    python tools/gpu/r04_pmc_kernels.py --summarise gpurun_out/<tag>/mfma_counter_collection.csv > profiles/r04_pmc_mfma_kernels.json
"""
import csv
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def summarise(path, names=("k_gemm_dw4", "k_attn_fwd", "k_attn_bwd", "Cijk", "attn_fwd", "bwd_kernel", "fmha")):
    per = {}
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r.get("Kernel_Name") or r.get("kernel_name")
            c = r.get("Counter_Name") or r.get("counter_name")
            v = float(r.get("Counter_Value") or r.get("counter_value") or 0)
            d = per.setdefault(k, {})
            e = d.setdefault(c, [0.0, 0])
            e[0] += v
            e[1] += 1
    out = {}
    for k, d in per.items():
        if not any(t in k for t in names):
            continue
        rec = {c: e[0] / e[1] for c, e in d.items()}
        rec["dispatches"] = max(e[1] for e in d.values())
        busy, gui = rec.get("SQ_VALU_MFMA_BUSY_CYCLES"), rec.get("GRBM_GUI_ACTIVE")
        if busy and gui:        # per-SIMD MFMA-busy cycles summed over the chip / (kernel cycles x 256 CUs x 4 SIMDs)
            rec["mfma_busy_fraction"] = busy / (gui * 256 * 4)
        wc = rec.get("SQ_WAVE_CYCLES")
        if wc:
            for c in ("SQ_WAIT_INST_LDS", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"):
                if c in rec:
                    rec[c + "_per_wave_cycle"] = rec[c] / wc
        out[k[:140]] = rec
    return out


def main():
    if len(sys.argv) > 2 and sys.argv[1] == "--summarise":
        print(json.dumps(summarise(sys.argv[2]), indent=1))
        return
    import torch

    from auto_round_amd import ops

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(0)
    bf = torch.bfloat16
    T = 16384
    for (M, N) in ((28672, 4096), (4096, 4096), (6144, 4096), (14336, 4096)):
        dY = (0.01 * torch.randn(T, M, device=dev, generator=g)).to(bf)
        X = torch.randn(T, N, device=dev, generator=g).to(bf)
        out = torch.empty(M, N, dtype=bf, device=dev)
        for _ in range(3):
            ops.gemm_dw(dY, X, out, split=False)
            torch.mm(dY.t(), X, out=out)
        del dY, X, out
    # attention: head size 128 (Llama-3-8B minibatch) forward; head size 64 (OPT-125M minibatch) forward + backward
    B, S, H, D = 8, 2048, 32, 128
    q, k, v = (torch.randn(B * S, H * D, device=dev, generator=g).to(bf) for _ in range(3))
    for _ in range(3):
        ops.attn_fwd(q, k, v, B, S, H, D)
    B, S, H, D = 8, 2048, 12, 64
    q, k, v, do = (torch.randn(B * S, H * D, device=dev, generator=g).to(bf) for _ in range(4))
    for _ in range(3):
        o, lse = ops.attn_fwd(q, k, v, B, S, H, D)
        ops.attn_bwd(q, k, v, o, lse, do, B, S, H, D)
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
