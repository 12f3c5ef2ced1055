"""Where does hipBLASLt's stream-K weight-gradient kernel cut its streamed tiles?  For every k-row s (multiples of `step`) the
MFMA kernel computes EVERY tile as [0, s) + [s, K) (ar_gemm_dw_sk) and the result is compared tile by tile with the library's:
a tile that matches at s (and not in one pass) is one the library sums in those two parts.  Writes the per-tile map."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16
gen = torch.Generator(device=DEV).manual_seed(2)
T = int(os.environ.get("AR_TOKENS", 16384))
step = int(sys.argv[1]) if len(sys.argv) > 1 else 32
shapes = dict(g=(14336, 4096), d=(4096, 14336), gu=(28672, 4096), qkv=(6144, 4096), o=(4096, 4096))
only = sys.argv[2].split(",") if len(sys.argv) > 2 else list(shapes)
out = {}
for name in only:
    o, i = shapes[name]
    dY = (0.01 * torch.randn(T, o, device=DEV, generator=gen)).to(BF)
    X = torch.randn(T, i, device=DEV, generator=gen).to(BF)
    lib = torch.mm(dY.t(), X).view(torch.int16)
    tm, tn = o // 256, i // 256
    tiles = tm * tn
    mine = torch.empty(o, i, dtype=BF, device=DEV)

    def tile_eq():
        return (mine.view(torch.int16) == lib).view(tm, 256, tn, 256).all(dim=3).all(dim=1).reshape(-1)

    ks = torch.zeros(tiles, dtype=torch.int32, device=DEV)
    assert ops.gemm_dw_sk(dY, X, mine, ks)
    one = tile_eq()
    ref1 = torch.empty_like(mine)
    assert ops.gemm_dw(dY, X, ref1, split=False)
    sane1 = bool(torch.equal(ref1.view(torch.int16), mine.view(torch.int16)))
    ks.fill_(T // 2)
    assert ops.gemm_dw_sk(dY, X, mine, ks)
    assert ops.gemm_dw(dY, X, ref1, split=2)
    sane2 = bool(torch.equal(ref1.view(torch.int16), mine.view(torch.int16)))
    found = torch.where(one, torch.zeros(tiles, dtype=torch.int64, device=DEV), torch.full((tiles,), -1, dtype=torch.int64, device=DEV))
    nmatch = one.to(torch.int64).clone()
    if not bool(one.all()):
        for s in range(step, T, step):
            ks.fill_(s)
            assert ops.gemm_dw_sk(dY, X, mine, ks)
            eq = tile_eq()
            nmatch += eq.to(torch.int64)
            found = torch.where((found < 0) & eq, torch.full_like(found, s), found)
    f = found.cpu().view(tm, tn)
    rec = dict(M=o, N=i, K=T, tiles=tiles, step=step, sane_one_pass=sane1, sane_two_slices=sane2, one_pass_tiles=int(one.sum()),
               two_part_tiles=int((f > 0).sum()), unmatched=int((f < 0).sum()), max_matches_per_tile=int(nmatch.max()),
               split_row_by_tile=f.tolist())
    out[name] = rec
    print(name, {k: v for k, v in rec.items() if k != "split_row_by_tile"}, flush=True)
os.makedirs("gpurun_out/r04t", exist_ok=True)
json.dump(out, open("gpurun_out/r04t/dw_streamk_probe.json", "w"))
