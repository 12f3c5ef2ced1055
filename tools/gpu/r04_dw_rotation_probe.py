"""Does hipBLASLt's weight-gradient kernel start its K loop at a per-tile offset (Tensile's StaggerU) at the 14336 x 4096 shapes?
For every rotation r of the token axis (multiples of `step` rows) the MFMA kernel's one-pass sum over the ROTATED operands is
compared, tile by tile, with the library's result on the unrotated ones: a tile that matches at rotation r is summed by the library
in the order r, r+1, ..., K-1, 0, ..., r-1."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(2)
T = 16384
step = int(sys.argv[1]) if len(sys.argv) > 1 else 32
out = {}
for name, (o, i) in dict(g=(14336, 4096), d=(4096, 14336)).items():
    dY = (0.01 * torch.randn(T, o, device=DEV, generator=g)).to(BF)
    X = torch.randn(T, i, device=DEV, generator=g).to(BF)
    lib = torch.mm(dY.t(), X).view(torch.int16)
    tm, tn = o // 256, i // 256
    found = torch.full((tm, tn), -1, dtype=torch.int64, device=DEV)
    mine = torch.empty(o, i, dtype=BF, device=DEV)
    for r in range(0, T, step):
        dYr = torch.roll(dY, -r, 0) if r else dY
        Xr = torch.roll(X, -r, 0) if r else X
        assert ops.gemm_dw(dYr, Xr, mine, split=False)
        eq = (mine.view(torch.int16) == lib).view(tm, 256, tn, 256).all(dim=3).all(dim=1)
        found = torch.where((found < 0) & eq, torch.full_like(found, r), found)
        if bool((found >= 0).all()):
            break
    f = found.cpu()
    rec = dict(tiles=tm * tn, matched=int((f >= 0).sum()), unmatched=int((f < 0).sum()), step=step,
               rotation_by_tile_row=[sorted(set(f[a].tolist())) for a in range(tm)][:64],
               rotation_grid_first_cols=[f[a, :8].tolist() for a in range(tm)][:64])
    out[name] = rec
    print(name, rec["matched"], rec["unmatched"], flush=True)
os.makedirs("gpurun_out/r04s", exist_ok=True)
json.dump(out, open("gpurun_out/r04s/dw_rotation_probe.json", "w"), indent=1)
print(json.dumps(out)[:3000])
