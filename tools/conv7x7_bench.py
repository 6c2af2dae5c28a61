"""gs_conv7x7_c4 at the bench shape (75 edges x 60 x 80) and the Replica / ScanNet map sizes: us per call over the
rows-per-workgroup choices, against the library path it replaces (MIOpen conv + gs_bias_act).  One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import go_slam_amd.droid_net as DN  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    conv = torch.nn.Conv2d(4, 128, 7, padding=3).to(dev)
    cache, hw = {}, DN._HalfWeights()
    out = {}
    for name, (n, h, w) in {"bench_75x60x80": (75, 60, 80), "replica_75x85x150": (75, 85, 150),
                            "scannet_75x48x64": (75, 48, 64)}.items():
        x = (4.0 * torch.randn(n, 4, h, w, device=dev)).half().contiguous(memory_format=torch.channels_last)
        r = {}
        from go_slam_amd import _lib
        L, st = _lib.lib(), _lib.stream_ptr(dev)
        DN.conv7x7_c4_bias_act(cache, conv, x, "relu")
        _, wp, bias = cache[id(conv)]
        y = torch.empty((n, 128, h, w), dtype=torch.float16, device=dev, memory_format=torch.channels_last)
        args = (_lib.ptr(x), _lib.ptr(wp), _lib.ptr(bias), _lib.ptr(y), 128, n, h, w, 1)
        for rt in (0, 2, 3, 4, 5, 6, 8, 10, 12, 15, 20):
            if rt > h:
                continue
            # bare C-ABI launches into one preallocated output: the host side is a few us per call
            r[f"rt{rt}"] = round(1e3 * bench.time_op(lambda: L.gs_conv7x7_c4(*args, rt, st), iters=200, warm=20), 1)
        r["wrapper_rt0"] = round(1e3 * bench.time_op(lambda: DN.conv7x7_c4_bias_act(cache, conv, x, "relu"), iters=50, warm=5), 1)
        r["library"] = round(1e3 * bench.time_op(lambda: DN.conv_bias_act(hw, conv, x, "relu"), iters=30, warm=5), 1)
        px = n * h * w
        r["hbm_floor_us"] = round(px * 264 / 8e12 * 1e6, 1)
        r["best_GBps"] = round(px * 264 / (min(v for k, v in r.items() if k.startswith("rt")) * 1e-6) / 1e9, 0)
        out[name] = r
    print(json.dumps(out))


if __name__ == "__main__":
    main()
