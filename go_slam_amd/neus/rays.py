"""Ray generation for the mapper (reference src/nerf_func.py:115-181 `build_rays`): random pixel
pick (mask-aware) and ray directions d = K^-1 [u, v, 1] R^T, o = t.  Negligible cost
(SURVEY.md 8 a11): stays PyTorch, RNG stays on the host side of the kernels."""
import numpy as np
import torch


def build_rays(H0, H1, W0, W1, n_rays, H, W, fx, fy, cx, cy, c2w, depth, color, device,
               nerf_coordinate=True, dir_normalize=False, mask=None):
    depth = depth[H0:H1, W0:W1]
    color = color[H0:H1, W0:W1]
    x, y = torch.meshgrid(torch.linspace(W0, W1 - 1, W1 - W0).to(device),
                          torch.linspace(H0, H1 - 1, H1 - H0).to(device), indexing="ij")
    x, y = x.t().reshape(-1), y.t().reshape(-1)
    depth = depth.reshape(-1)
    color = color.reshape(-1, 3)
    if mask is not None:
        keep = torch.nonzero(mask[H0:H1, W0:W1].reshape(-1).bool()).reshape(-1)
        x, y, depth, color = x[keep], y[keep], depth[keep], color[keep]
    N = x.shape[0]
    if 0 < n_rays < N // 2:
        idx = torch.randint(N, (n_rays,), device=device).clamp(0, N - 1)
        x, y, depth, color = x[idx], y[idx], depth[idx], color[idx]
    if isinstance(c2w, np.ndarray):
        c2w = torch.from_numpy(c2w).to(device)
    if nerf_coordinate:
        dirs = torch.stack([(x - cx) / fx, -(y - cy) / fy, -torch.ones_like(x)], dim=-1).to(device)
    else:
        dirs = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x)], dim=-1).to(device)
    if dir_normalize:
        raise TypeError("Ray direction shouldn't be normalized, otherwise the scale of pose will be destroyed!")
    rays_d = dirs @ c2w[:3, :3].t()
    rays_o = c2w[:3, 3].reshape(1, 3).repeat(x.shape[0], 1)
    return rays_o, rays_d, depth, color
