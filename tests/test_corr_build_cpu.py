"""CPU checks of the correlation-volume builder's index arithmetic (go_slam_amd/csrc/corr_build.hip) through its NumPy
restatement tools/emulate_corr_build.py.  The numerics on hardware: tests/test_track_gpu.py, test_benchshape_gpu.py."""
import importlib.util
import os

import numpy as np


def _emu():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emulate_corr_build.py")
    spec = importlib.util.spec_from_file_location("emulate_corr_build", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_fragment_order_is_a_bijection_and_one_load_is_one_kilobyte():
    E = _emu()
    for h, w in ((8, 8), (12, 16), (60, 80), (30, 40), (9, 24), (40, 60), (9, 12), (10, 92)):
        P = E.padded_pixels(h, w)
        assert P % 32 == 0 and P >= h * w and P >= (h + 3) // 4 * 4 * w and P >= (h * w + 63) // 64 * 64
        # the last n-tile of the last target-row block stays inside the padded image (w % 8 == 4: it is half full and
        # reads 16 pixels past the block)
        assert ((h + 3) // 4 * 4 - 4) * w + (4 * w + 31) // 32 * 32 <= P
        p, c = np.meshgrid(np.arange(P), np.arange(128), indexing="ij")
        off = E.frag_offset(p, c).ravel()
        assert off.min() == 0 and off.max() == P * 128 - 1 and np.unique(off).size == off.size
    # lane l of k-step ks of tile t reads the 8 halfs at ((t*8 + ks)*64 + l)*8: pixel 32 t + (l & 31), channels 16 ks + 8 (l >> 5) ..
    t, ks, lane = 3, 5, 41
    base = ((t * 8 + ks) * 64 + lane) * 8
    assert [E.frag_offset(32 * t + (lane & 31), 16 * ks + 8 * (lane >> 5) + k) for k in range(8)] == list(range(base, base + 8))


def test_xcd_renumbering_covers_every_tile_once_and_keeps_an_xcd_on_consecutive_tiles():
    E = _emu()
    for ntiles in (1, 7, 8, 9, 1125, 13500):
        nblocks = (ntiles + 7) // 8 * 8
        tiles = np.array([E.xcd_tile(b, nblocks) for b in range(nblocks)])
        live = tiles[tiles < ntiles]
        assert np.array_equal(np.sort(live), np.arange(ntiles))
        for xcd in range(8):                       # launch index mod 8 = XCD: its tiles are one contiguous run
            mine = tiles[xcd::8]
            assert np.array_equal(mine, np.arange(mine[0], mine[0] + mine.size))


def test_tile8_addresses_of_four_row_workgroups_fill_the_plane():
    """levels 0 and 1 in the tile8 layout: a workgroup writes the upper or lower 4 rows of a tile row (level 0) / 2 rows
    of level 1; over all workgroups every element of the (8-row padded) plane inside the map is written exactly once, and
    4 consecutive lanes of the level-0 store are 64 contiguous bytes"""
    E = _emu()
    for h, w in ((60, 80), (16, 16), (12, 32)):
        ntx = w >> 3
        plane = ntx * ((h + 7) >> 3) * 64
        seen = np.zeros(plane, np.int32)
        for y2_0 in range(0, h, E.ROWS):
            rows_valid = min(E.ROWS, h - y2_0)
            ybase = y2_0 & 7
            vec = ntx * E.ROWS
            for d in range(vec):
                tx, y = d // E.ROWS, d % E.ROWS
                if y < rows_valid:
                    a = ((y2_0 >> 3) * ntx + tx) * 64 + 8 * (ybase + y)
                    assert a == E.tile8_addr(y2_0 + y, 8 * tx, w)
                    seen[a:a + 8] += 1
            a0 = ((y2_0 >> 3) * ntx) * 64 + 8 * ybase
            assert [((y2_0 >> 3) * ntx + (d // 4)) * 64 + 8 * (ybase + d % 4) for d in range(4)] == [a0 + 8 * i for i in range(4)]
        inside = np.zeros(plane, bool)
        for y in range(h):
            for x in range(w):
                inside[E.tile8_addr(y, x, w)] = True
        assert np.array_equal(seen[inside], np.ones(inside.sum(), np.int32)) and seen[~inside].sum() == 0


def test_workgroup_pipeline_matches_a_plain_matmul():
    """fragment-ordered operands -> lane-level MFMA -> the LDS tile: c0[m][j] = <f1[:, p1_0 + m], f2[:, y2_0 * w + j]> / 16
    rounded to fp16 (the operands are scaled by 1/4 each, corr.py:71-72)"""
    E = _emu()
    rng = np.random.default_rng(5)
    h, w = 8, 16
    f1 = rng.standard_normal((128, h * w)).astype(np.float16)
    f2 = rng.standard_normal((128, h * w)).astype(np.float16)
    i1, i2 = E.prep(f1, h, w), E.prep(f2, h, w)
    for p1_0, y2_0 in ((0, 0), (64, 4)):
        c0 = E.volume_tile(i1, i2, h, w, p1_0, y2_0)
        a = (f1.T.astype(np.float16) / np.float16(4)).astype(np.float32)[p1_0:p1_0 + 64]
        b = (f2.T.astype(np.float16) / np.float16(4)).astype(np.float32)[y2_0 * w:(y2_0 + 4) * w]
        ref = (a @ b.T).astype(np.float16)
        assert np.allclose(c0.astype(np.float32), ref.astype(np.float32), rtol=2e-3, atol=2e-3)


def test_workgroup_pipeline_at_widths_with_a_half_full_column_tile():
    """w % 8 == 4 (EuRoC's 40 x 60 maps; here 8 x 12 and 12 x 20): a block of four target rows is an odd number of
    16-pixel halves, so every second block starts in the MIDDLE of a stored 32-pixel fragment tile and ends in a half-full
    MFMA column tile -- the per-lane fragment address of corr_volume_kernel's `frag` must pick the right pixels."""
    E = _emu()
    rng = np.random.default_rng(7)
    for h, w in ((8, 12), (12, 20)):
        f1 = rng.standard_normal((128, h * w)).astype(np.float16)
        f2 = rng.standard_normal((128, h * w)).astype(np.float16)
        i1, i2 = E.prep(f1, h, w), E.prep(f2, h, w)
        for p1_0 in (0, 64):
            for y2_0 in range(0, h, 4):
                c0 = E.volume_tile(i1, i2, h, w, p1_0, y2_0)
                m_valid = min(64, h * w - p1_0)
                a = (f1.T.astype(np.float16) / np.float16(4)).astype(np.float32)[p1_0:p1_0 + m_valid]
                b = (f2.T.astype(np.float16) / np.float16(4)).astype(np.float32)[y2_0 * w:(y2_0 + 4) * w]
                ref = (a @ b.T).astype(np.float16)
                assert c0.shape == (64, 4 * w)
                assert np.allclose(c0[:m_valid, :ref.shape[1]].astype(np.float32), ref.astype(np.float32), rtol=2e-3, atol=2e-3), (h, w, p1_0, y2_0)
