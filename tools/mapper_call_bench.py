"""Wall time of the reference's mapping loop on the substitutes: `Mapper.__call__` (src/mapping.py:151-302) on 20 filtered
keyframes of 480 x 640, mapping.pixels 4400, window 16, iters 20 (configs/go_slam.yaml's mapping block) -- with the
per-iteration ray draw through neus/rays.RayBank (default) and through build_rays frame by frame (the reference's form).
Prints one JSON line: ms per joint iteration for both."""
import json
import os
import sys
import time
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_slam_amd.neus as N                                   # noqa: E402
from go_slam_amd.depth_video import DepthVideo                  # noqa: E402
from go_slam_amd.neus import mapping as M                       # noqa: E402

dev = "cuda:0"
H, W, n_kf, iters = 480, 640, 20, 20
cfg = {"mode": "rgbd", "cam": {"H_out": H, "W_out": W}, "tracking": {"buffer": 32},
       "mapping": {"device": dev, "iters": iters, "decay": 0.5, "w_color_loss": 2.0, "w_sdf_loss": 2.0, "w_eikonal_loss": 0.1,
                   "uncertainty_weight_loss": True, "BA": False, "BA_cam_lr": 1e-3, "pixels": 4400, "mapping_window_size": 16,
                   "net_lr": 1e-3, "grid_lr": 1e-2}}
args = types.SimpleNamespace(device=dev)


def run(use_bank):
    torch.manual_seed(3)
    np.random.seed(3)
    video = DepthVideo.from_config(cfg, args)
    g = torch.Generator().manual_seed(7)
    v, u = torch.meshgrid(torch.arange(float(H)), torch.arange(float(W)), indexing="ij")
    depth = (2.0 + 0.2 * torch.sin(u * 0.02) * torch.cos(v * 0.03)).to(dev)
    video.images[:n_kf] = torch.rand(n_kf, 3, H, W, generator=g).to(dev)
    video.disps_filtered[:n_kf] = 1.0 / depth
    video.mask_filtered[:n_kf] = (torch.rand(n_kf, H, W, generator=g) < 0.9).float().to(dev)
    video.poses_filtered[:n_kf, 0] = 0.02 * torch.arange(n_kf, device=dev)
    video.update_priority[:n_kf] = 1.0
    video.timestamp[:n_kf] = torch.arange(n_kf, device=dev).float()
    video.bound[0] = torch.tensor([[-2.4, 2.4], [-2.4, 2.4], [-0.4, 2.4]], device=dev)
    video.filtered_id[0] = n_kf
    model = N.InstantNeuS({}, [[-2.5, 2.5]] * 3, device=dev).to(dev)
    slam = types.SimpleNamespace(verbose=False, bound=model.bound, video=video, mapping_net=model,
                                 renderer=N.Renderer(N_samples=24, N_surface=48), reload_map=torch.zeros(1).int(), H=H, W=W,
                                 fx=577.6, fy=578.7, cx=318.9, cy=242.7)
    mapper = M.Mapper(cfg, args, slam)
    mapper.use_ray_bank = use_bank
    for _ in range(4):                                       # the first call (10 x iterations on the new keyframes) and three
        mapper()                                             # steady ones: every ray-batch shape of the window has its graph
    torch.cuda.synchronize()
    g0 = mapper.global_step
    t = time.perf_counter()
    for _ in range(3):                                       # steady calls: `iters` joint iterations on the window each
        mapper()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    return 1e3 * dt / max(1, mapper.global_step - g0), mapper.global_step - g0


a = run(True)
b = run(False)
print(json.dumps({"workload": f"Mapper.__call__, {n_kf} keyframes of {H}x{W}, 4400 rays, window 16",
                  "ms_per_joint_iteration_ray_bank": round(a[0], 3), "ms_per_joint_iteration_build_rays_per_frame": round(b[0], 3),
                  "iterations_timed": [a[1], b[1]]}))
