"""Multi-view depth consistency filter + scene bound (mirrors src/multiview_filter.py:9-173):
hands filtered disparities, validity masks, poses and the scene bound from the tracker to the mapper.

The reference moves the full-resolution point cloud, counts and disparities of ALL keyframes to the host on every
call (`.cpu()` x3, ~40 B/pixel/keyframe over PCIe; multiview_filter.py:110-118) and compacts them there with boolean
indexing.  Here everything stays in HBM: the bound is a masked min / max over the device point cloud, the in-bound
test runs densely on the mask, and the only host traffic is the two `< 100 points` early-out scalars.  Results are
the same tensors (the reductions are exact).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import droid_backends
from .frontend import keyframe_count
from .lietorch_shim import SE3


def quat_to_euler(T):
    """[n, 7] (t, q xyzw) -> [n, 6] (t, roll, pitch, yaw)"""
    tx, ty, tz, x, y, z, w = torch.unbind(T, dim=-1)
    roll = torch.atan2(2.0 * (w * x + y * z), 1.0 - 2.0 * (x * x + y * y))
    pitch = torch.asin(torch.clamp(2.0 * (w * y - z * x), min=-1.0, max=1.0))
    yaw = torch.atan2(2.0 * (w * z + x * y), 1.0 - 2.0 * (y * y + z * z))
    return torch.stack([tx, ty, tz, roll, pitch, yaw], dim=-1)


def pose_dist(T0, T1):
    """BundleFusion-style pose change: |dt|_1 + 2 |d euler|_1 (src/multiview_filter.py:27-61)"""
    d = (quat_to_euler(T0) - quat_to_euler(T1)).abs()
    return d[:, :3].sum(dim=-1) + 2.0 * d[:, 3:].sum(dim=-1)


def masked_bound(points, mask, enlarge_scale=1.0):
    """[3, 2] (min, max) of points[mask] without compacting them (src/multiview_filter.py:82-97).
    points [..., 3], mask [...] bool, at least one True."""
    p = points.reshape(-1, 3)
    m = mask.reshape(-1, 1)
    inf = torch.tensor(float("inf"), device=p.device, dtype=p.dtype)
    lo = torch.where(m, p, inf).amin(dim=0)
    hi = torch.where(m, p, -inf).amax(dim=0)
    grow = (hi - lo) * (enlarge_scale - 1.0)
    return torch.stack([lo - grow / 2.0, hi + grow / 2.0], dim=-1)


def in_bound(points, bound):
    """strictly inside the box, per point (src/multiview_filter.py:63-79)"""
    b = bound.to(points.device)
    return ((points > b[:, 0]) & (points < b[:, 1])).all(dim=-1)


class MultiviewFilter(nn.Module):
    def __init__(self, cfg, args, slam):
        super().__init__()
        self.args, self.cfg = args, cfg
        self.device = args.device
        self.warmup = cfg["tracking"]["warmup"]
        mv = cfg["tracking"]["multiview_filter"]
        self.filter_thresh = mv["thresh"]                  # depth error bound
        self.filter_visible_num = mv["visible_num"]        # seen consistently by at least this many views
        self.kernel_size = mv["kernel_size"]
        self.bound_enlarge_scale = mv["bound_enlarge_scale"]
        self.net, self.video = slam.net, slam.video
        self.verbose = getattr(slam, "verbose", False)
        self.mode = getattr(slam, "mode", None)

    pose_dist = staticmethod(pose_dist)
    in_bound = staticmethod(in_bound)

    @staticmethod
    def get_bound_from_pointcloud(pts, enlarge_scale=1.0):
        return masked_bound(pts, torch.ones(pts.shape[:-1], dtype=torch.bool, device=pts.device), enlarge_scale)

    def _dilate(self, masks):
        if isinstance(self.kernel_size, str) and self.kernel_size == "inf":
            return torch.ones_like(masks)
        if int(self.kernel_size) < 2:
            return masks
        k = (int(self.kernel_size) // 2) * 2 + 1           # odd
        box = torch.ones(1, 1, k, k, dtype=torch.float32, device=masks.device)
        return F.conv2d(masks.unsqueeze(1).float(), box, padding=k // 2).bool().squeeze(1)

    @torch.no_grad()
    def forward(self):
        v = self.video
        cur_t = keyframe_count(v)
        filtered_t = int(v.filtered_id.item())
        if not (filtered_t < cur_t and cur_t > self.warmup):
            return
        with v.get_lock():
            index = torch.arange(cur_t, device=self.device)
            poses = v.poses[:cur_t].detach().clone()
            disps = v.disps_up[:cur_t].detach().clone()
            intrinsic = v.intrinsics[0].detach() * getattr(v, "scale_factor", 8)
            w2w = SE3(v.pose_compensate[0].clone().unsqueeze(0)).to(self.device)
        points = droid_backends.iproj((w2w * SE3(poses).inv()).data.contiguous(), disps, intrinsic.contiguous())
        thresh = self.filter_thresh * torch.ones_like(disps.mean(dim=[1, 2]))
        count = droid_backends.depth_filter(poses, disps, intrinsic.contiguous(), index, thresh)    # [b, h, w]
        masks = (count >= self.filter_visible_num) & (disps > 0.01 * disps.mean(dim=[1, 2], keepdim=True))
        if int(masks.sum()) < 100:
            return
        bound = masked_bound(points, masks)
        extended = self._dilate(masks)
        if int(extended.sum()) < 100:
            return
        extended = extended & in_bound(points, bound)      # dilated pixels must still fall inside the strict bound
        bound = masked_bound(points, extended)
        priority = pose_dist(v.poses_filtered[:cur_t].detach(), poses)
        lock = v.mapping.get_lock() if hasattr(getattr(v, "mapping", None), "get_lock") else v.get_lock()
        with lock:
            v.update_priority[:cur_t] += priority
            v.mask_filtered[:cur_t] = extended
            v.disps_filtered[:cur_t] = disps
            v.poses_filtered[:cur_t] = poses
            v.filtered_id[0] = cur_t
            v.bound[0] = bound
        if self.verbose:
            bd = bound.tolist()
            print(f"Multiview filtering: previous at {filtered_t}, now at {cur_t}; bound "
                  + ", ".join(f"[{a:.1f}, {b:.1f}]" for a, b in bd))
