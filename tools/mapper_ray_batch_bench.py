"""The mapper's per-iteration ray draw (src/mapping.py:222-240: build_rays once per visited keyframe and iteration) against
neus/rays.RayBank (frames stacked once per Mapper call), 16 keyframes of 480 x 640 with ragged masks, 275 rays each =
the reference's 4400-ray batch (configs/go_slam.yaml: mapping.pixels 4400, window 16).  Prints one JSON line:
wall time per draw with the device idle before and after (what a joint iteration adds around the 0.8 ms mapper step)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd.neus import rays as R      # noqa: E402

dev = "cuda:0"
H, W, F, n_rays = 480, 640, 16, 275
g = torch.Generator().manual_seed(3)
items = {}
for f in range(F):
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    c2w[:3, 3] = torch.randn(3, generator=g)
    items[f] = (torch.rand(H, W, 3, generator=g).to(dev), (torch.rand(H, W, generator=g) * 3 + 0.5).to(dev), c2w.to(dev), None,
                (torch.rand(H, W, generator=g) < 0.8).float().to(dev))
frames = list(range(F))
intr = (577.6, 578.7, 318.9, 242.7)


def per_frame():
    parts = [[], [], [], []]
    for f in frames:
        color, depth, c2w, _, mask = items[f]
        out = R.build_rays(0, H, 0, W, n_rays, H, W, *intr, c2w, depth, color, dev, nerf_coordinate=False, mask=mask)
        for acc, x in zip(parts, out):
            acc.append(x.float())
    return [torch.cat(p, 0) for p in parts]


def timed(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t) / iters


t0 = time.perf_counter()
bank = R.RayBank(items, H, W, *intr, dev)
torch.cuda.synchronize()
build_ms = 1e3 * (time.perf_counter() - t0)
torch.manual_seed(1)
a = per_frame()
torch.manual_seed(1)
b = bank.sample(frames, n_rays)
same = bool(torch.equal(a[0], b[0]) and torch.equal(a[3], b[2]) and torch.equal(a[2], b[3]) and
            torch.allclose(a[1], b[1], rtol=1e-5, atol=1e-6))
print(json.dumps({"workload": f"{F} keyframes x {H}x{W}, {n_rays} rays each", "per_frame_build_rays_ms": round(timed(per_frame), 3),
                  "ray_bank_sample_ms": round(timed(lambda: bank.sample(frames, n_rays)), 3),
                  "ray_bank_build_ms_once_per_mapper_call": round(build_ms, 3), "same_pixels": same}))
