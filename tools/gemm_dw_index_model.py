#!/usr/bin/env python
"""Index model of the weight-gradient GEMM kernels of csrc/ar_gemm.hip (CPU): k_gemm_dw4 (v_mfma_f32_32x32x16_bf16) and k_gemm_dw6
(v_mfma_f32_16x16x32_bf16).  The LDS image of a K-step of 32 (two units of 16 k-rows, each an X piece and a dY piece of [16][256]
bf16) is filled the way the kernels' LDS-DMA lanes address it, then every transposing fragment read (ds_read_b64_tr_b16) of every wave
is replayed with the hardware's rule -- within a 16-lane group, lane L receives element L % 4 of the 8-byte pieces supplied by lanes
L / 4, L / 4 + 4, L / 4 + 8, L / 4 + 12 (profiles/archive/r02_mfma_probe.json) -- and checked:
  * every lane gets, for its MFMA operand, the (column, 8 consecutive k) the instruction's layout wants
        32x32x16: column lane % 32 of the fragment, k = 8 (lane / 32) + j;   16x16x32: column lane % 16, k = 8 (lane / 16) + j;
  * every (k, column) of both pieces is read exactly as often as the wave grid implies;
  * each half-wave of a read (ds_read_b64_tr_b16 is served as 2 x 32 lanes, 64 banks of 4 bytes) touches 32 distinct 8-byte slots;
  * every output element of the 256 x 256 tile is stored exactly once.
The formulas are restated from the kernels, so a change on one side only fails tests/test_gemm_dw_index_model.py.
"""
import numpy as np

ROWB, PIECE, UNIT = 512, 8192, 16384


def swz(row, m16):
    """XOR on the 16-byte chunk index of k-row `row` (0..15) of a piece"""
    return ((row & 3) << 2) | ((((row >> 3) & 1) << 1) if m16 else 0)


def stage_pair(m16):
    """LDS image of one K-step of 32 as tags (operand, k, column) per bf16 element, filled as the DMA lanes address it:
    wave w stages k-rows 2w, 2w + 1 of each piece; lane -> row 2w + (lane >> 5), PHYSICAL chunk lane & 31, SOURCE chunk (lane & 31) ^ swz"""
    lds = np.full((2 * UNIT // 2, 3), -1, dtype=np.int64)
    for unit in range(2):
        for op in (0, 1):                       # 0: X piece (P), 1: dY piece (Q)
            for wave in range(8):
                for lane in range(64):
                    drow = 2 * wave + (lane >> 5)
                    pc = lane & 31
                    lchunk = pc ^ swz(drow, m16)
                    dst = unit * UNIT + op * PIECE + 2 * wave * ROWB + lane * 16          # lane-linear destination
                    assert dst == unit * UNIT + op * PIECE + drow * ROWB + pc * 16
                    for e in range(8):
                        lds[dst // 2 + e] = (op, unit * 16 + drow, lchunk * 8 + e)
    return lds


def tr_read(lds, addrs):
    """ds_read_b64_tr_b16 of one wave: addrs[lane] = byte address of the lane's 8-byte piece -> [64][4] tags"""
    out = np.zeros((64, 4, 3), dtype=np.int64)
    for lane in range(64):
        grp, L = lane >> 4, lane & 15
        for t in range(4):
            sup = 16 * grp + (L // 4) + 4 * t
            out[lane, t] = lds[addrs[sup] // 2 + (L % 4)]
    return out


def frag_addrs(m16, wave, op, frag, hi, old_swizzle=False):
    """byte addresses (inside the pair) the 64 lanes of `wave` supply for fragment `frag` of operand `op` (0: X / P, 1: dY / Q);
    old_swizzle: the 16x16x32 read pattern on the 32x32x16 kernel's swizzle (the negative control of the test)"""
    wm, wn = wave & 3, wave >> 2
    a = np.zeros(64, dtype=np.int64)
    for lane in range(64):
        q, i = lane >> 4, lane & 15
        rowsel, piece = i >> 2, i & 3
        if m16:
            s = (rowsel << 2) | (0 if old_swizzle else ((q & 1) << 1))
            rowoff = (q >> 1) * UNIT + (8 * (q & 1) + rowsel) * ROWB + (piece & 1) * 8
            chunk = (wn * 16 + frag * 2 + (piece >> 1)) if op == 0 else (wm * 8 + frag * 2 + (piece >> 1))
            a[lane] = op * PIECE + rowoff + ((chunk ^ s) << 4) + (4 * ROWB if hi else 0)
        else:
            g = q >> 1
            rowoff = (8 * g + rowsel) * ROWB + (piece & 1) * 8
            chunk = (wn * 16 + frag * 4 + (q & 1) * 2 + (piece >> 1)) if op == 0 else (wm * 8 + frag * 4 + (q & 1) * 2 + (piece >> 1))
            a[lane] = op * PIECE + rowoff + ((chunk ^ (rowsel << 2)) << 4) + (4 * ROWB if hi else 0)
    return a


def worst_slot_multiplicity(addrs):
    """distinct addresses per 8-byte slot of the 256-byte bank row, per half-wave"""
    worst = 0
    for half in (0, 1):
        slots = {}
        for lane in range(32 * half, 32 * half + 32):
            slots.setdefault((int(addrs[lane]) // 8) % 32, set()).add(int(addrs[lane]))
        worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def check(m16):
    lds = stage_pair(m16)
    worst = 0
    reads = {0: {}, 1: {}}
    nfrag = {0: 8 if m16 else 4, 1: 4 if m16 else 2}
    width = 16 if m16 else 32
    for wave in range(8):
        wm, wn = wave & 3, wave >> 2
        for op in (0, 1):
            base_col = wn * 128 if op == 0 else wm * 64
            for frag in range(nfrag[op]):
                units = (0,) if m16 else (0, 1)          # the 32x32x16 kernel reads the pair's two units with the same addresses + UNIT
                for unit in units:
                    for hi in (0, 1):
                        addrs = frag_addrs(m16, wave, op, frag, hi) + (0 if m16 else unit * UNIT)
                        worst = max(worst, worst_slot_multiplicity(addrs))
                        got = tr_read(lds, addrs)
                        for lane in range(64):
                            col = base_col + frag * width + (lane % width)
                            k0 = (8 * (lane // 16)) if m16 else (unit * 16 + 8 * (lane // 32))
                            for t in range(4):
                                want = (op, k0 + 4 * hi + t, col)
                                assert tuple(got[lane, t]) == want, (m16, wave, op, frag, unit, hi, lane, t, tuple(got[lane, t]), want)
                                reads[op][want[1:]] = reads[op].get(want[1:], 0) + 1
    # every (k, column) of the X piece is read by the 4 waves sharing a wn (wm = 0..3), of the dY piece by the 2 sharing a wm
    assert len(reads[0]) == 32 * 256 and set(reads[0].values()) == {4}
    assert len(reads[1]) == 32 * 256 and set(reads[1].values()) == {2}
    return worst


def epilogue_cover(m16):
    hit = np.zeros((256, 256), dtype=np.int32)
    for wave in range(8):
        wm, wn = wave & 3, wave >> 2
        for lane in range(64):
            if m16:         # acc[mi][ni][r]: m = wm*64 + mi*16 + lane % 16, n = wn*128 + ni*16 + 4 (lane / 16) + r
                for mi in range(4):
                    for ni in range(8):
                        for r in range(4):
                            hit[wm * 64 + mi * 16 + (lane & 15), wn * 128 + ni * 16 + 4 * (lane >> 4) + r] += 1
            else:           # acc[mi][ni][4 t + r]: m = wm*64 + mi*32 + lane % 32, n = wn*128 + ni*32 + 8 t + 4 (lane / 32) + r
                for mi in range(2):
                    for ni in range(4):
                        for t in range(4):
                            for r in range(4):
                                hit[wm * 64 + mi * 32 + (lane & 31), wn * 128 + ni * 32 + 8 * t + 4 * (lane >> 5) + r] += 1
    return bool((hit == 1).all())


def worst_16x16x32_on_the_old_swizzle():
    """why the swizzle gained bit 3 of the k-row: the two 16-lane groups of a half-wave read the same columns 8 k-rows apart"""
    return max(worst_slot_multiplicity(frag_addrs(True, w, op, f, 0, old_swizzle=True)) for w in range(8) for op in (0, 1) for f in range(4))


def main():
    out = {}
    for m16 in (False, True):
        out["16x16x32" if m16 else "32x32x16"] = (check(m16), epilogue_cover(m16))
    print("; ".join(f"{k}: operands ok, worst distinct addresses per 8-byte bank slot within a half-wave {w}, epilogue covers the tile once: {e}"
                    for k, (w, e) in out.items()))
    return out


if __name__ == "__main__":
    main()
