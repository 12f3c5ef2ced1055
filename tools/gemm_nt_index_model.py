#!/usr/bin/env python
"""Index model of csrc/ar_gemm_nt.hip (CPU): the LDS image as an array of tags, the LDS-DMA fills as the kernel's lanes address them,
then every fragment read of every wave -- each lane must receive exactly the (row, 8 consecutive k) its MFMA operand wants, every
(row, chunk) of a stage must be read, and every 16-lane group of a ds_read_b128 must cover 16 distinct 16-byte bank slots.

What it pins (the formulas are restated here from the kernel, so a change on one side only fails tests/test_gemm_nt_index_model.py):
  * LDS layout: operand op in {A, B}, half h, stage parity sg at byte  op * 65536 + h * 32768 + sg * 16384, rows of 128 bytes;
  * DMA piece j of wave (wr, wc): rows r = wc * 32 + 8 j + (lane >> 3) of half wr, lane -> physical chunk lane & 7 holding the row's
    LOGICAL chunk (lane & 7) ^ ((r >> 1) & 7) -- the swizzle lives on the global SOURCE address, the destination is lane-linear;
  * fragment read of K half a, k16 unit u: lane (l31 = lane & 31, h = lane >> 5) reads row l31 of its 32-row tile at physical chunk
    (4 a + 2 u + h) ^ ((l31 >> 1) & 7); tiles are immediate offsets (mi * 4096 / ni * 4096), the stage parity another (sg * 16384).
"""
import numpy as np


def halfbase(op, h, sg):
    return op * 65536 + h * 32768 + sg * 16384


def dma_stage(lds, t, sg):
    """all 8 waves issue their A pieces and B pieces (both of half wr) of stage t into parity sg"""
    for wave in range(8):
        wr, wc = wave >> 2, wave & 3
        for op in (0, 1):
            for j in range(4):
                dst = halfbase(op, wr, sg) + (wc * 32 + 8 * j) * 128
                for lane in range(64):
                    r = wc * 32 + 8 * j + (lane >> 3)
                    lc = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7)            # the kernel's form of (r >> 1) & 7
                    assert ((r >> 1) & 7) == ((4 * j + (lane >> 4)) & 7)
                    lds[(dst + lane * 16) // 16] = (op, wr * 128 + r, t * 8 + lc)


def frag_offset(lane, a, u):
    l31, h = lane & 31, lane >> 5
    return l31 * 128 + 16 * ((4 * a + 2 * u + h) ^ ((l31 >> 1) & 7))


def check_reads(lds, t, sg):
    seen_a, seen_b = set(), set()
    for wave in range(8):
        wr, wc = wave >> 2, wave & 3
        for a in (0, 1):
            for u in (0, 1):
                for lane in range(64):
                    l31, h = lane & 31, lane >> 5
                    x = frag_offset(lane, a, u)
                    for mi in range(4):
                        op, row, ch = lds[(halfbase(0, wr, 0) + x + sg * 16384 + mi * 4096) // 16]
                        assert (op, row, ch) == (0, wr * 128 + mi * 32 + l31, t * 8 + 4 * a + 2 * u + h)
                        seen_a.add((row, ch))
                    for ni in range(2):
                        op, row, ch = lds[(65536 + (wc >> 1) * 32768 + (wc & 1) * 8192 + x + sg * 16384 + ni * 4096) // 16]
                        assert (op, row, ch) == (1, wc * 64 + ni * 32 + l31, t * 8 + 4 * a + 2 * u + h)
                        seen_b.add((row, ch))
    assert len(seen_a) == 256 * 8 and len(seen_b) == 256 * 8


def worst_bank_multiplicity():
    """ds_read_b128 is served in four 16-lane groups (MI355X_MICROARCH.md, LDS table); a 16-byte slot of the 256-byte bank row hit by
    more than one distinct address within a group is a conflict"""
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    worst = 0
    for a in (0, 1):
        for u in (0, 1):
            for g in groups:
                slots = {}
                for lane in g:
                    x = frag_offset(lane, a, u)
                    slots.setdefault((x // 16) % 16, set()).add(x)
                worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def epilogue_cover():
    """every (m, n) of the 256 x 256 tile is stored exactly once: lane owns row wr*128 + mi*32 + (lane & 31), columns
    wc*64 + ni*32 + 8 t + 4 (lane >> 5) + r (accumulator register 4 t + r of acc[mi][ni])"""
    hit = np.zeros((256, 256), dtype=np.int32)
    for wave in range(8):
        wr, wc = wave >> 2, wave & 3
        for lane in range(64):
            for mi in range(4):
                for ni in range(2):
                    for t in range(4):
                        for r in range(4):
                            hit[wr * 128 + mi * 32 + (lane & 31), wc * 64 + ni * 32 + 8 * t + 4 * (lane >> 5) + r] += 1
    return bool((hit == 1).all())


# ---- nt2 (one wave per SIMD, 128 x 128 per wave): same LDS image; wave w stages rows [64 w, 64 w + 64) of each operand tile (8 pieces),
# reads A rows wr * 128 + mi * 32 + l31 (wr = w >> 1) and B rows wc * 128 + ni * 32 + l31 (wc = w & 1); unit u of a stage = chunks 2u, 2u+1
def dma_stage_nt2(lds, t, sg):
    for wave in range(4):
        for op in (0, 1):
            for j in range(8):
                dst = op * 65536 + (wave >> 1) * 32768 + sg * 16384 + ((wave & 1) * 64 + 8 * j) * 128
                for lane in range(64):
                    row = wave * 64 + 8 * j + (lane >> 3)
                    lc = (lane & 7) ^ ((4 * j + (lane >> 4)) & 7)
                    assert (((row & 127) >> 1) & 7) == ((4 * j + (lane >> 4)) & 7)
                    lds[(dst + lane * 16) // 16] = (op, row, t * 8 + lc)


def check_reads_nt2(lds, t, sg):
    seen_a, seen_b = set(), set()
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        for u in range(4):
            for lane in range(64):
                l31, h = lane & 31, lane >> 5
                x = l31 * 128 + 16 * ((2 * u + h) ^ ((l31 >> 1) & 7))
                for i in range(4):
                    op, row, ch = lds[(wr * 32768 + x + sg * 16384 + i * 4096) // 16]
                    assert (op, row, ch) == (0, wr * 128 + i * 32 + l31, t * 8 + 2 * u + h)
                    seen_a.add((row, ch))
                    op, row, ch = lds[(65536 + wc * 32768 + x + sg * 16384 + i * 4096) // 16]
                    assert (op, row, ch) == (1, wc * 128 + i * 32 + l31, t * 8 + 2 * u + h)
                    seen_b.add((row, ch))
    assert len(seen_a) == 256 * 8 and len(seen_b) == 256 * 8


def epilogue_cover_nt2():
    hit = np.zeros((256, 256), dtype=np.int32)
    for wave in range(4):
        wr, wc = wave >> 1, wave & 1
        for lane in range(64):
            for mi in range(4):
                for ni in range(4):
                    for t in range(4):
                        for r in range(4):
                            hit[wr * 128 + mi * 32 + (lane & 31), wc * 128 + ni * 32 + 8 * t + 4 * (lane >> 5) + r] += 1
    return bool((hit == 1).all())


# ---- the 16x16x32 form (k_gemm_nt<.., M16 = true>): same LDS image and DMA; fragment f (16 rows) of K half a: lane (l15 = lane & 15,
# kg = lane >> 4) reads row 16 f + l15 at physical chunk (4 a + kg) ^ (l15 >> 1); fragments are immediate offsets f * 2048
def frag_offset_m16(lane, a):
    l15, kg = lane & 15, lane >> 4
    return l15 * 128 + 16 * ((4 * a + kg) ^ (l15 >> 1))


def check_reads_m16(lds, t, sg):
    seen_a, seen_b = set(), set()
    for wave in range(8):
        wr, wc = wave >> 2, wave & 3
        for a in (0, 1):
            for lane in range(64):
                l15, kg = lane & 15, lane >> 4
                x = frag_offset_m16(lane, a)
                for f in range(8):
                    op, row, ch = lds[(halfbase(0, wr, 0) + x + sg * 16384 + f * 2048) // 16]
                    assert (op, row, ch) == (0, wr * 128 + f * 16 + l15, t * 8 + 4 * a + kg)
                    seen_a.add((row, ch))
                for f in range(4):
                    op, row, ch = lds[(65536 + (wc >> 1) * 32768 + (wc & 1) * 8192 + x + sg * 16384 + f * 2048) // 16]
                    assert (op, row, ch) == (1, wc * 64 + f * 16 + l15, t * 8 + 4 * a + kg)
                    seen_b.add((row, ch))
    assert len(seen_a) == 256 * 8 and len(seen_b) == 256 * 8


def worst_bank_multiplicity_m16():
    groups = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
    groups += [[x + 32 for x in g] for g in groups]
    worst = 0
    for a in (0, 1):
        for g in groups:
            slots = {}
            for lane in g:
                x = frag_offset_m16(lane, a)
                slots.setdefault((x // 16) % 16, set()).add(x)
            worst = max(worst, max(len(v) for v in slots.values()))
    return worst


def epilogue_cover_m16():
    """lane owns row wr*128 + mi*16 + lane % 16, columns wc*64 + ni*16 + 4 (lane / 16) + r (register r of acc16[mi][ni])"""
    hit = np.zeros((256, 256), dtype=np.int32)
    for wave in range(8):
        wr, wc = wave >> 2, wave & 3
        for lane in range(64):
            for mi in range(8):
                for ni in range(4):
                    for r in range(4):
                        hit[wr * 128 + mi * 16 + (lane & 15), wc * 64 + ni * 16 + 4 * (lane >> 4) + r] += 1
    return bool((hit == 1).all())


def main_m16():
    lds = np.full((131072 // 16, 3), -1, dtype=np.int64)
    dma_stage(lds, 0, 0)
    dma_stage(lds, 1, 1)
    check_reads_m16(lds, 0, 0)
    check_reads_m16(lds, 1, 1)
    return worst_bank_multiplicity_m16(), epilogue_cover_m16()


def main_nt2():
    lds = np.full((131072 // 16, 3), -1, dtype=np.int64)
    dma_stage_nt2(lds, 0, 0)
    dma_stage_nt2(lds, 1, 1)
    check_reads_nt2(lds, 0, 0)
    check_reads_nt2(lds, 1, 1)
    dma_stage_nt2(lds, 2, 0)
    check_reads_nt2(lds, 2, 0)
    check_reads_nt2(lds, 1, 1)
    return epilogue_cover_nt2()


def main():
    lds = np.full((131072 // 16, 3), -1, dtype=np.int64)
    dma_stage(lds, 0, 0)
    dma_stage(lds, 1, 1)
    check_reads(lds, 0, 0)
    check_reads(lds, 1, 1)
    dma_stage(lds, 2, 0)
    check_reads(lds, 2, 0)
    check_reads(lds, 1, 1)          # restaging parity 0 left parity 1 alone
    w = worst_bank_multiplicity()
    ok = epilogue_cover()
    ok2 = main_nt2()
    w16, ok16 = main_m16()
    print(f"mapping ok; worst distinct addresses per 16-byte bank slot within a lane group: {w}; epilogue covers the tile exactly once: {ok}; "
          f"nt2 mapping ok, epilogue: {ok2}; 16x16x32 form: mapping ok, worst {w16}, epilogue: {ok16}")
    return max(w, w16), ok and ok2 and ok16


if __name__ == "__main__":
    main()
