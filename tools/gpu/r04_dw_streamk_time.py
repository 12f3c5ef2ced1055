"""The stream-K structure found on one gradient pair reproduces the library on another; what each form of the GEMM costs."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops, streamk  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16
T = 16384
out = {}


def timed(fn, n=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


for name, (o, i) in dict(g=(14336, 4096), d=(4096, 14336)).items():
    gen = torch.Generator(device=DEV).manual_seed(11)
    dY = (0.01 * torch.randn(T, o, device=DEV, generator=gen)).to(BF)
    X = torch.randn(T, i, device=DEV, generator=gen).to(BF)
    st = streamk.find_on_device(dY, X)
    rec = dict(found=st is not None)
    if st is not None:
        s, kc = st
        if s is not None:
            rec.update(grid=s.grid, wgm=s.wgm, depth=s.depth, one_pass_tiles=s.n_dp, two_part_tiles=s.two_part_tiles)
        else:
            rec.update(one_pass=True)
        same = []
        for seed in (12, 13, 14):
            gen.manual_seed(seed)
            dY2 = (0.02 * torch.randn(T, o, device=DEV, generator=gen)).to(BF)
            X2 = (torch.randn(T, i, device=DEV, generator=gen) * 3).to(BF)
            lib = torch.mm(dY2.t(), X2)
            mine = torch.empty_like(lib)
            assert ops.gemm_dw_sk(dY2, X2, mine, kc)
            same.append(bool(torch.equal(lib.view(torch.int16), mine.view(torch.int16))))
        rec["equal_on_other_operands"] = same
        mine = torch.empty(o, i, dtype=BF, device=DEV)
        fl = 2.0 * T * o * i
        for label, fn in dict(library=lambda: torch.mm(dY.t(), X, out=mine), mfma_own_plan=lambda: ops.gemm_dw(dY, X, mine),
                              mfma_one_pass=lambda: ops.gemm_dw(dY, X, mine, split=False),
                              mfma_streamk_structure=lambda: ops.gemm_dw_sk(dY, X, mine, kc)).items():
            ms = timed(fn)
            rec[label] = dict(ms=round(ms, 4), pflops=round(fl / ms / 1e12, 3))
    out[name] = rec
    print(name, json.dumps(rec), flush=True)
# gate | up as ONE launch: every layer's rows in the structure of its own library GEMM (streamk.find_merged_on_device)
o, i = 14336, 4096
gen = torch.Generator(device=DEV).manual_seed(31)
dGU = (0.01 * torch.randn(T, 2 * o, device=DEV, generator=gen)).to(BF)
X = torch.randn(T, i, device=DEV, generator=gen).to(BF)
kc = streamk.find_merged_on_device(dGU, X, [o, o])
rec = dict(found=kc is not None)
if kc is not None:
    sep = torch.cat([torch.mm(dGU[:, :o].contiguous().t(), X), torch.mm(dGU[:, o:].contiguous().t(), X)])
    mine = torch.empty(2 * o, i, dtype=BF, device=DEV)
    assert ops.gemm_dw_sk(dGU, X, mine, kc)
    rec["equal_to_the_two_library_gemms"] = bool(torch.equal(sep.view(torch.int16), mine.view(torch.int16)))
    dg, du = dGU[:, :o].contiguous(), dGU[:, o:].contiguous()
    kg = streamk.find_on_device(dg, X)[1]
    for label, fn in dict(library_two_gemms=lambda: (torch.mm(dg.t(), X, out=mine[:o]), torch.mm(du.t(), X, out=mine[o:])),
                          streamk_two_launches=lambda: (ops.gemm_dw_sk(dg, X, mine[:o], kg), ops.gemm_dw_sk(du, X, mine[o:], kg)),
                          streamk_one_merged_launch=lambda: ops.gemm_dw_sk(dGU, X, mine, kc)).items():
        rec[label] = dict(ms=round(timed(fn), 4))
out["gate_up_merged"] = rec
print("gate_up_merged", json.dumps(rec), flush=True)
os.makedirs("gpurun_out/r04t", exist_ok=True)
json.dump(out, open("gpurun_out/r04t/dw_streamk_time.json", "w"), indent=1)
