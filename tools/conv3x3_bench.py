"""Micro-benchmark: the package's implicit-GEMM 3x3 convolution (gs_conv3x3_pp) vs MIOpen on the update operator's layer
shapes.  Prints one JSON object per map shape.  (The round-1 / round-2 A/B against the retired kernels:
profiles/r01_conv3x3_bench.json, profiles/r02_conv3x3_bench.json.)

    python tools/conv3x3_bench.py                 # bench workload: 75 edges, 60x80
    python tools/conv3x3_bench.py all             # + Replica (40x80) and ScanNet (30x40, 13-keyframe update_lowmem chunk)
"""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import droid_net as DN  # noqa: E402

LAYERS = (("gru_zr", 320, 256), ("gru_q", 320, 128), ("heads", 128, 384), ("corr_enc2", 128, 128))
SHAPES = {"S480": (75, 60, 80), "Rep": (75, 40, 80), "Scan": (78, 30, 40)}


def time_op(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def run(tag, E, h, w):
    dev = "cuda:0"
    out = {"shape": tag, "edges": E, "map": [h, w], "tile_width": DN.conv3x3_pp_tile_width(w)}
    for name, c, o in LAYERS:
        x = torch.randn(E, c, h, w, device=dev).half().contiguous(memory_format=torch.channels_last)
        wt = (torch.randn(o, c, 3, 3, device=dev) / (3 * c ** 0.5)).half().contiguous(memory_format=torch.channels_last)
        flops = 2.0 * E * h * w * 9 * c * o
        ms_m = time_op(lambda: F.conv2d(x, wt, None, padding=1))
        row = {"miopen_ms": round(ms_m, 4), "miopen_tflops": round(flops / ms_m / 1e9, 1)}
        ref = F.conv2d(x, wt, None, padding=1).float()
        ms_p = time_op(lambda: DN.conv3x3_hip(x, wt))
        row["pp_ms"] = round(ms_p, 4)
        row["pp_tflops"] = round(flops / ms_p / 1e9, 1)
        row["pp_max_abs_diff"] = float((DN.conv3x3_hip(x, wt).float() - ref).abs().max())
        out[name] = row
    print(json.dumps(out))


if __name__ == "__main__":
    torch.backends.cudnn.benchmark = True
    for tag in (SHAPES if len(sys.argv) > 1 and sys.argv[1] == "all" else ("S480",)):
        run(tag, *SHAPES[tag])
