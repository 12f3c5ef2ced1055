import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as orc
from auto_round_amd import ops
z = np.load("tests/golden/int_qdq_w4g128_sym_f16.npz")
nbits, gs, sym, rows, cols = [int(x) for x in z["meta"]]
W = orc.from_bits(z["W"].reshape(-1), orc.DT_F16).cuda()
for trial in range(3):
    wmin, wmax = ops.group_minmax(W, gs)
    Wq, scale, zp = ops.qdq_int_fwd(W, torch.from_numpy(z["V"].reshape(-1)).cuda(), wmin, wmax, torch.from_numpy(z["min_scale"]).cuda(),
                                    torch.from_numpy(z["max_scale"]).cuda(), gs=gs, bits=nbits, sym=sym, want_scale=True)
    a = orc.to_bits(Wq); b = z["Wq"].reshape(-1)
    bad = np.nonzero(a != b)[0]
    print("trial", trial, "mismatch", len(bad), bad[:10], [hex(x) for x in a[bad[:6]]], [hex(x) for x in b[bad[:6]]], "groups", np.unique(bad // gs)[:10])
    # with the golden wmin/wmax
    Wq2 = ops.qdq_int_fwd(W, torch.from_numpy(z["V"].reshape(-1)).cuda(), orc.from_bits(z["wmin"], orc.DT_F16).cuda(), orc.from_bits(z["wmax"], orc.DT_F16).cuda(),
                          torch.from_numpy(z["min_scale"]).cuda(), torch.from_numpy(z["max_scale"]).cuda(), gs=gs, bits=nbits, sym=sym)
    print("   with golden wmin/wmax mismatch", int((orc.to_bits(Wq2) != b).sum()))
    print("   wmin/wmax g5", hex(orc.to_bits(wmin)[5]), hex(orc.to_bits(wmax)[5]), hex(z["wmin"][5]), hex(z["wmax"][5]))
