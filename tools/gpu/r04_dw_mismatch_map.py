"""Where does the library's weight-gradient GEMM (torch.mm(dY.t(), X), hipBLASLt) differ from the MFMA kernel's one-pass result at
Llama-3-8B's 14336 x 4096 shapes?  Counts per 256 x 256 output tile, and both results against an fp64 reference on a sample of tiles."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from auto_round_amd import ops  # noqa: E402

DEV, BF = "cuda:0", torch.bfloat16
g = torch.Generator(device=DEV).manual_seed(2)
T = 16384
out = {}
for name, (o, i) in dict(g=(14336, 4096), d=(4096, 14336)).items():
    dY = (0.01 * torch.randn(T, o, device=DEV, generator=g)).to(BF)
    X = torch.randn(T, i, device=DEV, generator=g).to(BF)
    lib = torch.mm(dY.t(), X)
    mine = torch.empty_like(lib)
    assert ops.gemm_dw(dY, X, mine, split=False)
    ne = (lib.view(torch.int16) != mine.view(torch.int16))
    per_tile = ne.view(o // 256, 256, i // 256, 256).sum(dim=(1, 3))
    nz = per_tile.nonzero().tolist()
    rec = dict(total=int(ne.sum()), tiles_with_mismatch=len(nz), tiles=(o // 256) * (i // 256),
               worst=[(a, b, int(per_tile[a, b])) for a, b in sorted(nz, key=lambda t: -int(per_tile[t[0], t[1]]))[:12]],
               rows_hist=[int(v) for v in ne.sum(dim=1).view(-1, 256).sum(dim=1).tolist()][:64],
               cols_hist=[int(v) for v in ne.sum(dim=0).view(-1, 256).sum(dim=1).tolist()][:64])
    # within the worst tile: which rows / columns
    if nz:
        a, b, _ = rec["worst"][0]
        sub = ne[a * 256:(a + 1) * 256, b * 256:(b + 1) * 256]
        rec["worst_tile_rows"] = [int(v) for v in sub.sum(dim=1).tolist()]
        rec["worst_tile_cols"] = [int(v) for v in sub.sum(dim=0).tolist()]
        # accuracy of both against fp64 on that tile
        ref = (dY[:, a * 256:(a + 1) * 256].double().t() @ X[:, b * 256:(b + 1) * 256].double())
        rec["err_lib_vs_fp64"] = float((lib[a * 256:(a + 1) * 256, b * 256:(b + 1) * 256].double() - ref).abs().max())
        rec["err_mine_vs_fp64"] = float((mine[a * 256:(a + 1) * 256, b * 256:(b + 1) * 256].double() - ref).abs().max())
        rec["ref_absmax"] = float(ref.abs().max())
    out[name] = rec
    del dY, X, lib, mine
print(json.dumps(out))
os.makedirs("gpurun_out/r04d", exist_ok=True)
json.dump(out, open("gpurun_out/r04d/dw_mismatch_map.json", "w"), indent=1)
