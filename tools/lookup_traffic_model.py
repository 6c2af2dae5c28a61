"""How many 128-byte lines does one correlation lookup touch, per volume layout?  (CPU model, NumPy only.)

The fused lookup (csrc/corr_lookup.hip) reads, per source pixel and pyramid level, the 8 x 8 texels around its target
coordinate from that source pixel's OWN correlation plane (src/modules/corr.py:43-53, radius 3 + bilinear).  The PMC
passes (profiles/r03_pmc_corr_lookup.json) show ~2.4 x more bytes fetched than the 128 B per level the algorithm
needs, because an 8 x 8 window rarely sits on line boundaries.  This tool counts the distinct lines per window for
candidate layouts on a synthetic flow field of the bench's shape, so that the next layout is chosen by numbers:

  rowmajor        [src][y][x]
  tile8           8 x 8 tiles (levels 0-1; the production layout), levels 2-3 row-major
  tile4x16        4 rows x 16 columns per line (what a 4-row builder workgroup writes as WHOLE lines)
  src2x2_tile4    the planes of a 2 x 2 block of source pixels interleaved: one line = 4 x 4 texels x 4 sources; neighbouring
                  source pixels look at almost the same window, so a line fetched once serves up to four lookups
  src1x4_tile4    the same with 4 consecutive source pixels of a row

    python tools/lookup_traffic_model.py [--edges 75] [--h 60] [--w 80] [--flow 12] > profiles/r03_lookup_layout_model.json
"""
import argparse
import json

import numpy as np

LINE = 64            # fp16 elements per 128-byte line


def smooth_flow(n, h, w, amp, rng):
    """per-edge flow field: a global shift + a low-frequency warp, like camera motion over a mostly rigid scene"""
    ys, xs = np.meshgrid(np.arange(h), np.arange(w), indexing="ij")
    out = np.empty((n, h, w, 2), np.float32)
    for e in range(n):
        shift = rng.uniform(-amp, amp, 2)
        a = rng.uniform(-0.08, 0.08, (2, 2))
        ph = rng.uniform(0, 6.28, 2)
        fx = shift[0] + a[0, 0] * (xs - w / 2) + a[0, 1] * (ys - h / 2) + 1.5 * np.sin(xs / 11.0 + ph[0])
        fy = shift[1] + a[1, 0] * (xs - w / 2) + a[1, 1] * (ys - h / 2) + 1.5 * np.cos(ys / 9.0 + ph[1])
        out[e, ..., 0], out[e, ..., 1] = xs + fx, ys + fy
    return out


def window_taps(coords, level, hl, wl):
    """integer texel coordinates [N, 64] (x, y) and validity of the 8 x 8 window at this level"""
    c = coords.reshape(-1, 2) / float(1 << level)
    x0 = np.floor(c[:, 0]).astype(np.int64) - 3
    y0 = np.floor(c[:, 1]).astype(np.int64) - 3
    dx, dy = np.meshgrid(np.arange(8), np.arange(8), indexing="xy")
    tx = x0[:, None] + dx.reshape(1, -1)
    ty = y0[:, None] + dy.reshape(1, -1)
    ok = (tx >= 0) & (tx < wl) & (ty >= 0) & (ty < hl)
    return tx, ty, ok


def distinct(ids, ok):
    """number of distinct non-negative ids per row (invalid entries ignored)"""
    ids = np.where(ok, ids, -1)
    s = np.sort(ids, axis=1)
    first = np.concatenate([s[:, :1] >= 0, (s[:, 1:] != s[:, :-1]) & (s[:, 1:] >= 0)], axis=1)
    return first.sum(axis=1)


def lines_per_lookup(layout, coords, h, w):
    """mean number of 128-byte lines fetched per source pixel, per level, for one layout"""
    n = coords.shape[0]
    res = []
    src = np.arange(n * h * w, dtype=np.int64)                        # global source-pixel (plane) index
    sy, sx = (src % (h * w)) // w, (src % (h * w)) % w
    for level in range(4):
        hl, wl = h >> level, w >> level
        tx, ty, ok = window_taps(coords, level, hl, wl)
        kind = layout if (level <= 1 or layout in ("rowmajor",)) else "rowmajor"      # levels 2-3 stay row-major planes
        if kind == "rowmajor":
            plane = hl * wl
            addr = src[:, None] * plane + ty * wl + tx
            res.append(float(distinct(addr // LINE, ok).mean()))
        elif kind in ("tile8", "tile4x16"):
            th, tw = (8, 8) if kind == "tile8" else (4, 16)
            ntx, nty = -(-wl // tw), -(-hl // th)
            line = src[:, None] * (ntx * nty) + (ty // th) * ntx + (tx // tw)
            res.append(float(distinct(line, ok).mean()))
        else:                                                          # source-interleaved: groups of 4 planes share lines
            if kind == "src2x2_tile4":
                gid = (src // (h * w)) * ((h // 2) * (w // 2)) + (sy // 2) * (w // 2) + (sx // 2)
            else:
                gid = src // 4
            ntx, nty = -(-wl // 4), -(-hl // 4)
            line = gid[:, None] * (ntx * nty) + (ty // 4) * ntx + (tx // 4)
            order = np.argsort(gid, kind="stable")
            g_line = line[order].reshape(-1, 4 * 64)
            g_ok = ok[order].reshape(-1, 4 * 64)
            res.append(float(distinct(g_line, g_ok).sum() / line.shape[0]))      # lines per group / 4 sources
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--edges", type=int, default=75)
    ap.add_argument("--h", type=int, default=60)
    ap.add_argument("--w", type=int, default=80)
    ap.add_argument("--flow", type=float, default=12.0)
    a = ap.parse_args()
    rng = np.random.default_rng(7)
    coords = smooth_flow(a.edges, a.h, a.w, a.flow, rng)
    out = {"shape": {"edges": a.edges, "h": a.h, "w": a.w, "flow_amplitude_px": a.flow},
           "algorithmic_lines_per_lookup": 4.0, "layouts": {}}
    for layout in ("rowmajor", "tile8", "tile4x16", "src2x2_tile4", "src1x4_tile4"):
        per_level = lines_per_lookup(layout, coords, a.h, a.w)
        total = sum(per_level)
        out["layouts"][layout] = {"lines_per_level": [round(v, 3) for v in per_level], "lines_per_lookup": round(total, 3),
                                  "overfetch_vs_512B": round(total * 128 / 512.0, 3),
                                  "read_MB_per_update": round(total * 128 * a.edges * a.h * a.w / 1e6, 1)}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
