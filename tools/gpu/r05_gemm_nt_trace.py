"""Per-phase cycle table of the NT GEMM (csrc/ar_gemm_nt.hip with s_memtime bookkeeping, ar_gemm_nt_trace): where a wave's cycles go --
fragment reads, parked at the barrier after them, the 16-MFMA cluster (with or without the LDS-DMA issues in it), parked at the barrier
after it -- for both DMA-issue variants, on Llama-3-8B's o-projection forward shape (16384 x 4096 x 4096, random operands).  VERDICT r04
item 2: "... or a per-phase cycle table (s_memtime or PMC) showing where the remaining 40 % of the pipe goes".

    python tools/gpu/r05_gemm_nt_trace.py --out gpurun_out/r05/gemm_nt_phase_cycles.json
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import _lib  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=None)
    ap.add_argument("--variants", default="0,1,3", help="0 / 1: 32x32x16 with the DMA between the MFMAs / at the end of the read part; 3: 16x16x32")
    args = ap.parse_args()
    lib = _lib.load()
    g = torch.Generator(device="cuda").manual_seed(0)
    out = dict(device=torch.cuda.get_device_name(0), cases=[])
    for (M, N, K) in [(16384, 4096, 4096), (16384, 4096, 14336)]:
        A = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
        B = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
        C = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
        tiles = (M // 256) * (N // 256)
        for variant in [int(v) for v in args.variants.split(",")]:
            tr = torch.zeros((tiles, 8, 8), dtype=torch.int64, device="cuda")
            st = torch.cuda.current_stream().cuda_stream
            for _ in range(2):
                rc = lib.ar_gemm_nt_trace(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N, tr.data_ptr(), variant, st)
                assert rc == 0, rc
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            lib.ar_gemm_nt_trace(A.data_ptr(), B.data_ptr(), C.data_ptr(), M, N, K, K, K, N, tr.data_ptr(), variant, st)
            e1.record()
            e1.synchronize()
            ms = e0.elapsed_time(e1)
            t = tr.double().cpu()
            phases = float(t[0, 0, 5])
            per = t[:, :, :5] / phases                        # cycles per phase, [tile, wave, segment]
            grp = {"group0_waves_0_3": per[:, :4].mean(dim=(0, 1)).tolist(), "group1_waves_4_7": per[:, 4:].mean(dim=(0, 1)).tolist()}
            allw = per.mean(dim=(0, 1)).tolist()
            clock_ghz = float((t[:, :, 4] / t[:, :, 6].clamp(min=1)).mean()) * 0.1          # shader cycles per 100 MHz tick
            rec = dict(M=M, N=N, K=K, variant=variant, dma_issue={0: "32x32x16, DMA between the MFMAs (variant 0)", 1: "32x32x16, DMA at the end of the read part of even phases (variant 1)",
                                                   3: "16x16x32, DMA between the MFMAs (variant 3)"}[variant],
                       ms_traced=ms, pflops_traced=2.0 * M * N * K / ms / 1e12, phases=int(phases),
                       cycles_per_phase=dict(read_part=allw[0], barrier_after_reads=allw[1], mfma_part=allw[2], barrier_after_mfma=allw[3], whole_phase=allw[4]),
                       by_group=grp, mfma_cycles_if_back_to_back=16 * 32, shader_clock_ghz_during_the_k_loop=clock_ghz,
                       pflops_at_this_clock_if_the_pipe_never_idled=256 * 4 * 1024 * clock_ghz * 1e9 / 1e15,
                       mfma_pipe_busy_fraction_of_a_simd=2 * 16 * 32 / allw[4],
                       note="a SIMD hosts one wave of each group; per phase each issues 16 MFMAs of 32 cycles (variant 3: 32 of 16): the pipe is busy 1024 cycles of "
                            "every `whole_phase` cycles a wave takes")
            print(json.dumps(rec), flush=True)
            out["cases"].append(rec)
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
