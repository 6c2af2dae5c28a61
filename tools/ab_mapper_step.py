"""Wall time of the fused mapper step (hipGraph replay) at 4096 and 32768 rays, plus a checksum of the trained state, for
same-box A/B runs of a knob or a second library build:
    GOSLAM_FORK_BIN_REDUCE=0 python tools/ab_mapper_step.py ; GOSLAM_FORK_BIN_REDUCE=1 python tools/ab_mapper_step.py
Prints one JSON line: ms per step (median of 5 blocks of `iters` steps) and an exact checksum of the table after the run
(two variants that launch the same kernels in a different order must print the same checksum)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_slam_amd.neus as neus                     # noqa: E402
from go_slam_amd.neus.mapper import MapTrainer      # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dev = torch.device("cuda:0")
out = {"fork": os.environ.get("GOSLAM_FORK_BIN_REDUCE", "1"), "lib": os.environ.get("GOSLAM_HIP_LIB", "")}
for n in (4096, 32768):
    g = torch.Generator().manual_seed(43)
    model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(dev)
    with torch.no_grad():
        p = model.sdf_network.encoding.encoding.params
        p.copy_((torch.rand(p.shape, generator=g) - 0.5) * 0.02)
        model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(dev) * 0.1
    R = neus.Renderer(N_samples=24, N_surface=48)
    o = (torch.rand(n, 3, generator=g) * 6 - 3).to(dev)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(dev)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[torch.rand(n, generator=g) < 0.1] = 0
    gt = gt.to(dev)
    col = torch.rand(n, 3, generator=g).to(dev)
    pr = torch.rand(24, generator=g).to(dev)
    tr = MapTrainer(model, R)
    for _ in range(6):
        loss = tr.step(o, d, col, gt, pr)
    torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(iters):
            loss = tr.step(o, d, col, gt, pr)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / iters * 1e3)
    ts.sort()
    tab = model.sdf_network.encoding.encoding.params.detach().double()
    out[str(n)] = {"ms_per_step": round(ts[2], 4), "min": round(ts[0], 4), "loss": float(loss),
                   "table_sum": float(tab.sum()), "table_abs_sum": float(tab.abs().sum())}
print(json.dumps(out))
