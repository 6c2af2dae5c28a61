"""Path-M legs of bench.py as a stand-alone command for rocprofv3 (kernel trace or --pmc passes):
    python tools/profile_mapping.py render|train|train_weak [iters]
`render` = Renderer.render_batch_ray + InstantNeuS.forward on 4096 rays x 72 samples; `train` = one mapper iteration on
32768 rays; `train_weak` = the same on 4096 rays (the reference mapper's batch).  Every iteration does the same work
(warm-up included), so per-step numbers are trace totals / iters."""
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import go_slam_amd.neus as neus                     # noqa: E402
from go_slam_amd.neus.mapper import MapTrainer      # noqa: E402

leg = sys.argv[1] if len(sys.argv) > 1 else "render"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n = {"render": 4096, "train": 32768, "train_weak": 4096}[leg]
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(43)
model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(dev)
with torch.no_grad():
    p = model.sdf_network.encoding.encoding.params
    p.copy_((torch.rand(p.shape, generator=g) - 0.5) * (0.1 if leg == "render" else 0.02))
    model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(dev) * 0.1
R = neus.Renderer(N_samples=24, N_surface=48)
o = (torch.rand(n, 3, generator=g) * 6 - 3).to(dev)
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(dev)
gt = torch.rand(n, generator=g) * 3.5 + 0.5
gt[torch.rand(n, generator=g) < 0.1] = 0
gt = gt.to(dev)
col = torch.rand(n, 3, generator=g).to(dev)
pr = torch.rand(24, generator=g).to(dev)
if leg == "render":
    with torch.no_grad():
        for _ in range(iters):
            R.render_batch_ray(o, d, model, None, dev, gt)
else:
    tr = MapTrainer(model, R)
    for _ in range(iters):
        tr.step(o, d, col, gt, pr)
torch.cuda.synchronize()
print("done", leg, iters)
