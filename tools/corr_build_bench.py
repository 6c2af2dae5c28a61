"""gs_corr_volume_pyramid (all-pairs volume + 3 pooled levels) at the bench shape: ms per 8-edge launch, both layouts,
and the achieved write bandwidth.  One JSON line.  `--only NAME [--layout tile8|rowmajor]` restricts the run to one
shape / layout (what the --pmc passes and the kernel trace of profiles/r03_pmc_corr_volume.json ran)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from go_slam_amd import droid_backends as db  # noqa: E402


def main():
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default=None)
    ap.add_argument("--layout", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {}
    for name, (nb, ht, wd) in {"bench_8x60x80": (8, 60, 80), "scannet_8x48x64": (8, 48, 64), "half_8x30x40": (8, 30, 40),
                               "bench_25x60x80": (25, 60, 80)}.items():
        if args.only and name != args.only:
            continue
        f1 = torch.randn(nb, 128, ht, wd, device=dev).half()
        f2 = torch.randn(nb, 128, ht, wd, device=dev).half()
        hw = ht * wd
        nbytes = nb * (hw * hw * 2 * (1 + 1 / 4 + 1 / 16 + 1 / 64) + 2 * hw * 128 * 2)
        r = {}
        for lname, layout in (("rowmajor", db.CORR_ROWMAJOR), ("tile8", db.CORR_TILE8)):
            if (layout == db.CORR_TILE8 and not db.corr_tile8_supported(f1)) or (args.layout and lname != args.layout):
                continue
            ms = bench.time_op(lambda: db.corr_volume_pyramid(f1, f2, layout), iters=10, warm=3)
            r[lname] = {"ms": round(ms, 4), "GBps": round(nbytes / ms / 1e6, 0)}
        out[name] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
