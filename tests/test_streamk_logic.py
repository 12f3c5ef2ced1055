"""Host logic of the stream-K summation structure (auto_round_amd/streamk.py) against the map measured on the library
(tests/golden/streamk_llama8b_dw_map.json, made by tools/gpu/r04_dw_streamk_probe.py on an MI355X in round 4; the script is kept)."""
import json
import os

import numpy as np
import pytest

from auto_round_amd import streamk as sk

GOLD = os.path.join(os.path.dirname(__file__), "golden", "streamk_llama8b_dw_map.json")


def test_tile_order_is_a_permutation_and_walks_bands():
    for tm, tn, w in [(56, 16, 6), (16, 56, 6), (5, 7, 3), (4, 4, 1), (3, 9, -4), (8, 8, -8), (7, 3, 16)]:
        o = sk.tile_order(tm, tn, w)
        assert sorted(o.tolist()) == list(range(tm * tn))
    o = sk.tile_order(5, 3, 2).tolist()
    # band of rows 0-1: columns in turn, rows fastest; then rows 2-3; then the single row 4
    assert o == [0, 3, 1, 4, 2, 5, 6, 9, 7, 10, 8, 11, 12, 13, 14]
    assert sk.tile_order(2, 5, -2).tolist() == [0, 1, 5, 6, 2, 3, 7, 8, 4, 9]
    assert sk.tile_order(3, 4, 1).tolist() == list(range(12))
    with pytest.raises(ValueError):
        sk.tile_order(2, 2, 0)


def test_structure_counts():
    st = sk.structure(56, 16, 16384, 246, 6, 32)
    assert st.n_dp == 492 and len(st.ksplit) == 404 and st.n_dp % 246 == 0
    assert st.two_part_tiles == 245                       # 245 cuts, none on a tile boundary
    assert (st.ksplit % 32 == 0).all() and st.ksplit.max() < 16384
    # a grid that divides the tiles: everything in one pass except the streamed last `grid` tiles, which the cuts leave whole
    st = sk.structure(7, 32, 4096, 224, 4, 32)
    assert st.n_dp == 0 or st.two_part_tiles == 0
    # runs shorter than a tile would cut a tile twice: not a structure ar_gemm_dw_sk can sum
    assert sk.structure(2, 2, 16384, 16, 1, 32) is None
    assert sk.structure(4, 4, 16, 8, 1, 32) is None       # K below one iteration


@pytest.mark.parametrize("name", ["g", "d"])
def test_structure_reproduces_the_measured_map(name):
    rec = json.load(open(GOLD))[name]
    seen = np.array(rec["split_row_by_tile"])
    tm, tn = seen.shape
    st = sk.structure(tm, tn, rec["K"], 246, 6, 32)
    pred = np.zeros(tm * tn, dtype=np.int64)
    pred[st.tlist[st.n_dp:]] = st.ksplit
    pred = pred.reshape(tm, tn)
    differ = np.argwhere(pred != seen)
    # the only disagreements: cuts within 45 iterations of the tile end, where the probe's first matching s is not unique
    assert len(differ) <= 8
    for r, c in differ:
        assert pred[r, c] >= (512 - 45) * 32, (r, c, pred[r, c], seen[r, c])
    # every tile the probe found to be in two parts is a two-part tile of the structure
    assert ((seen > 0) <= (pred > 0)).all()


@pytest.mark.parametrize("name", ["g", "d"])
def test_discover_finds_the_structure_from_the_mismatch_mask(name):
    rec = json.load(open(GOLD))[name]
    seen = np.array(rec["split_row_by_tile"])
    found = sk.discover(seen > 0, rec["K"])
    assert found and (found[0].grid, found[0].wgm, found[0].depth, found[0].n_dp) == (246, 6, 32, 492)
    assert sk.discover(np.zeros_like(seen, dtype=bool), rec["K"]) == []


def test_discover_round_trip_on_synthetic_structures():
    rng = np.random.default_rng(0)
    for tm, tn, K, grid, wgm, depth in [(24, 16, 8192, 200, 8, 32), (16, 24, 8192, 120, -4, 64), (40, 8, 4096, 96, 1, 32)]:
        st = sk.structure(tm, tn, K, grid, wgm, depth)
        assert st is not None
        mask = np.zeros(tm * tn, dtype=bool)
        split = st.tlist[st.n_dp:][st.ksplit > 0]
        mask[split] = True
        hide = rng.choice(split, size=min(2, len(split)), replace=False)      # split tiles that happen to equal the one-pass sum
        mask[hide] = False
        found = sk.discover(mask.reshape(tm, tn), K)
        assert any(f.key() == st.key() for f in found), (tm, tn, grid, wgm, depth)


def test_merged_table_is_the_parts_tables_in_row_order(monkeypatch):
    import torch
    sk._merged.clear()
    seen = []

    def fake_find(dY, X, lib_out=None):
        seen.append((tuple(dY.shape), dY.is_contiguous()))
        m = dY.shape[1]
        if m == 768:
            return None
        return (None, torch.full(((m // 256) * (X.shape[1] // 256),), m, dtype=torch.int32))

    monkeypatch.setattr(sk, "find_on_device", fake_find)
    dY, X = torch.zeros(64, 512 + 256, dtype=torch.bfloat16), torch.zeros(64, 512, dtype=torch.bfloat16)
    kc = sk.find_merged_on_device(dY, X, [512, 256])
    assert kc.tolist() == [512] * 4 + [256] * 2                  # 2 x 2 tiles of the first part, then 1 x 2 of the second
    assert seen == [((64, 512), True), ((64, 256), True)]        # contiguous copies: the operand form of the module path's own call
    assert sk.find_merged_on_device(dY, X, [512, 256]) is kc     # cached
    assert sk.find_merged_on_device(dY, X, [256, 256]) is None   # rows that do not add up
    assert sk.find_merged_on_device(dY, X, [384, 384]) is None   # not whole tiles
    dY2 = torch.zeros(64, 768 + 256, dtype=torch.bfloat16)
    assert sk.find_merged_on_device(dY2, X, [768, 256]) is None  # a part without a reproducing structure
    sk._merged.clear()


def test_structure_invariants_over_random_problems():
    """what every structure must satisfy, whatever the library chose: the one-pass tiles are whole rounds of the grid, at most
    grid - 1 tiles are cut, cuts are multiples of the iteration depth strictly inside (0, K), the launch order is a permutation, and the
    row-major table is the launch-order table re-indexed"""
    rng = np.random.default_rng(7)
    seen = 0
    for _ in range(400):
        tm, tn = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        depth = int(rng.choice([32, 64, 128]))
        K = depth * int(rng.integers(1, 600))
        grid = int(rng.integers(1, 400))
        wgm = int(rng.choice(sk.WGMS))
        st = sk.structure(tm, tn, K, grid, wgm, depth)
        if st is None:
            continue
        seen += 1
        tiles = tm * tn
        assert sorted(st.tlist.tolist()) == list(range(tiles))
        assert len(st.ksplit) == tiles - st.n_dp
        if tiles > grid:
            assert st.n_dp % grid == 0 and grid <= tiles - st.n_dp < 2 * grid
        else:
            assert st.n_dp == 0
        assert st.two_part_tiles <= grid - 1 or grid == 1
        cuts = st.ksplit[st.ksplit > 0]
        assert ((cuts % depth) == 0).all() and (cuts < K).all()
        kc = st.kcut()
        assert kc.shape == (tiles,) and int((kc > 0).sum()) == st.two_part_tiles
        assert (kc[st.tlist[:st.n_dp]] == 0).all()
        assert (kc[st.tlist[st.n_dp:]] == st.ksplit).all()
    assert seen > 100
