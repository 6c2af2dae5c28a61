"""Ray generation for the mapper (reference src/nerf_func.py:115-181 `build_rays`): random pixel
pick (mask-aware) and ray directions d = K^-1 [u, v, 1] R^T, o = t.  Negligible cost
(SURVEY.md 8 a11): stays PyTorch, RNG stays on the host side of the kernels."""
import os

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


class RayBank:
    """The mapper's per-iteration ray draw without its per-frame glue.  `Mapper.__call__` (src/mapping.py:222-240,
    262-283) calls `build_rays` once per visited keyframe in EVERY joint iteration: per frame a pixel grid, a `nonzero`
    of the mask (a host sync), four full-resolution gathers, the random pick and the direction arithmetic -- ~25
    launches and a sync, x 16-22 frames = several milliseconds per iteration around a mapper step that takes 0.8 ms
    here.  What does not change between the iterations of one call (the frames' colours, depths, masks, poses) is
    stacked ONCE; the rank -> pixel map of each mask is its running sum, and the valid-pixel counts reach the host in
    one transfer per call.  A draw is then: the SAME `torch.randint(N_f, (n_rays,))` calls in the same frame order as
    `build_rays` (a seeded run picks the same pixels as the reference), one `searchsorted` over the concatenated
    running sums, two gathers and the direction arithmetic for all frames at once -- F + ~15 launches, no sync.
    Frames whose mask leaves fewer than 2 n_rays pixels take `build_rays` itself (it then returns EVERY valid pixel)."""

    def __init__(self, items, H, W, fx, fy, cx, cy, device):
        self.items, self.H, self.W, self.device = items, H, W, device
        self.intr = (fx, fy, cx, cy)
        self.frames = list(items.keys())
        self.pos = {f: i for i, f in enumerate(self.frames)}
        F, HW = len(self.frames), H * W
        if F == 0:
            self.N = []
            return
        color, depth, c2w, mask = [], [], [], []
        for f in self.frames:
            col, dep, pose, _, msk = items[f]
            color.append(col.reshape(HW, 3))
            depth.append(dep.reshape(HW))
            if isinstance(pose, np.ndarray):
                pose = torch.from_numpy(pose)
            c2w.append(pose.to(device))
            mask.append(torch.ones(HW, dtype=torch.bool, device=device) if msk is None else msk.reshape(HW).bool())
        self.color = torch.cat(color, 0)                                    # [F HW, 3]
        self.depth = torch.cat(depth, 0)                                    # [F HW]
        c2w = torch.stack(c2w, 0)
        self.rot_t = c2w[:, :3, :3].transpose(1, 2).contiguous()            # dirs @ R^T per frame
        self.trans = c2w[:, :3, 3].contiguous()
        cums = torch.stack(mask, 0).to(torch.int64).cumsum(1)               # rank of every valid pixel, 1-based
        self.N = [int(n) for n in cums[:, -1].tolist()]                     # the ONE host transfer of a Mapper call
        # the device draw (gs_ray_draw: one launch per iteration) wants fp32 planes it can index and 32-bit running sums
        self.fused = (os.environ.get("GOSLAM_RAY_DRAW_FUSED", "1") != "0" and self.color.is_cuda and self.color.dtype == torch.float32 and self.depth.dtype == torch.float32
                      and c2w.dtype == torch.float32 and HW < 2 ** 31)
        if self.fused:
            self.color, self.depth = self.color.contiguous(), self.depth.contiguous()
            self.cum32 = cums.to(torch.int32).contiguous()
            self._fpos = {}
            return
        # one sorted array for all frames: frame f's running sum shifted by f (HW + 1) -- a rank query of frame f
        # (1 .. N_f <= HW) cannot land in another frame's stretch
        self.big = HW + 1
        self.cums = (cums + self.big * torch.arange(F, device=cums.device)[:, None]).reshape(-1)

    def sample(self, frames, n_rays):
        """(rays_o, rays_d, color, depth) of `n_rays` random valid pixels of each of `frames`, concatenated in order"""
        H, W = self.H, self.W
        fx, fy, cx, cy = self.intr
        dev = self.device
        pos = [self.pos[f] for f in frames]
        if not pos or any(not (0 < n_rays < self.N[p] // 2) for p in pos):
            parts = [[], [], [], []]                                       # (rare: tiny masks -- the reference's own form)
            for f in frames:
                color, depth, c2w, _, mask = self.items[f]
                out = build_rays(0, H, 0, W, n_rays, H, W, fx, fy, cx, cy, c2w, depth, color, dev,
                                 nerf_coordinate=False, dir_normalize=False, mask=mask)
                for acc, x in zip(parts, out):
                    acc.append(x.float())
            rays_o, rays_d, depth, color = (torch.cat(p, dim=0) for p in parts)
            return rays_o, rays_d, color, depth
        if self.fused:
            return self._sample_fused(pos, n_rays)
        # the reference's random draws, call for call
        idx = torch.stack([torch.randint(self.N[p], (n_rays,), device=dev).clamp(0, self.N[p] - 1) for p in pos], 0)
        fi = torch.tensor(pos, dtype=torch.int64, device=dev)
        HW = H * W
        g = torch.searchsorted(self.cums, (idx + 1 + self.big * fi[:, None]).reshape(-1))     # global pixel index
        local = g - (HW * fi)[:, None].expand(-1, n_rays).reshape(-1)
        x = (local % W).float()
        y = torch.div(local, W, rounding_mode="floor").float()
        dirs = torch.stack([(x - cx) / fx, (y - cy) / fy, torch.ones_like(x)], dim=-1).reshape(len(pos), n_rays, 3)
        rays_d = torch.bmm(dirs, self.rot_t[fi].to(dirs.dtype)).reshape(-1, 3)
        rays_o = self.trans[fi].to(dirs.dtype)[:, None, :].expand(-1, n_rays, -1).reshape(-1, 3)
        return rays_o.float(), rays_d.float(), self.color[g].float(), self.depth[g].float()

    def _sample_fused(self, pos, n_rays):
        """The same draw on the device path: the reference's `torch.randint` calls, call for call (written into the rows of
        one index tensor; `clamp(0, N - 1)` is the identity on their range), then ONE launch (gs_ray_draw: rank -> pixel
        through the running mask sums, directions, origins, colour and depth of every frame's rays)."""
        from .. import _lib
        dev = self.color.device
        nf = len(pos)
        idx = torch.empty(nf, n_rays, dtype=torch.int64, device=dev)
        for k, p in enumerate(pos):
            torch.randint(self.N[p], (n_rays,), device=dev, out=idx[k])
        key = tuple(pos)
        fpos = self._fpos.get(key)
        if fpos is None:
            if len(self._fpos) > 64:
                self._fpos.clear()
            fpos = self._fpos[key] = torch.tensor(pos, dtype=torch.int32, device=dev)
        f32 = dict(dtype=torch.float32, device=dev)
        n = nf * n_rays
        rays_o, rays_d, color = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
        depth = torch.empty(n, **f32)
        fx, fy, cx, cy = self.intr
        with torch.cuda.device(dev):
            rc = _lib.lib().gs_ray_draw(_lib.ptr(idx), _lib.ptr(fpos), _lib.ptr(self.cum32), _lib.ptr(self.color),
                                        _lib.ptr(self.depth), _lib.ptr(self.rot_t), _lib.ptr(self.trans), nf, n_rays,
                                        self.H * self.W, self.W, float(fx), float(fy), float(cx), float(cy),
                                        _lib.ptr(rays_o), _lib.ptr(rays_d), _lib.ptr(color), _lib.ptr(depth),
                                        _lib.stream_ptr(dev))
        _lib.check(rc, "RayBank.sample")
        return rays_o, rays_d, color, depth
