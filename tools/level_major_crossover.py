"""Where the level-major gather order of gs_neus_forward starts to pay: Renderer.render_batch_ray + InstantNeuS.forward
(inference: no records for the backward) at 4096 ... 32768 rays x 72 samples in both orders (gs_neus_level_major_min_points
forced to 0 / 2^30).  Prints one JSON object (-> profiles/r05_level_major_crossover.json)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import _lib  # noqa: E402
import go_slam_amd.neus as neus  # noqa: E402

dev = torch.device("cuda:0")
L = _lib.lib()
g = torch.Generator().manual_seed(43)
model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(dev)
with torch.no_grad():
    p = model.sdf_network.encoding.encoding.params
    p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.1)
    model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(dev) * 0.1
R = neus.Renderer(N_samples=24, N_surface=48)
out = {"default_min_points": L.gs_neus_level_major_min_points(-1), "forward_us": {}}
for n in (4096, 6144, 8192, 12288, 16384, 24576, 32768):
    o = (torch.rand(n, 3, generator=g) * 6 - 3).to(dev)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(dev)
    gt = (torch.rand(n, generator=g) * 3.5 + 0.5).to(dev)
    row = {}
    with torch.no_grad():
        z, dist = R.sample(o, d, model.bound, gt)
        for name, thr in (("per_point", 1 << 30), ("level_major", 0)):
            old = L.gs_neus_level_major_min_points(thr)
            for _ in range(3):
                model(o, d, z, dist)
            torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(20):
                model(o, d, z, dist)
            b.record()
            b.synchronize()
            row[name] = round(a.elapsed_time(b) / 20 * 1e3, 1)
            L.gs_neus_level_major_min_points(old)
    out["forward_us"][str(n)] = dict(row, points=n * 72)
print(json.dumps(out))
