"""Poses of the non-keyframes (mirrors src/trajectory_filler.py:8-112): each batch of up to 16 frames is parked behind
the keyframes in the video buffers, initialised by constant-velocity interpolation between its two neighbouring
keyframes, tied to them with two edges per frame and refined by 6 motion-only updates (update operator + dense BA on
the HIP path).

The keyframe bracket of every timestamp is found with one comparison matrix on the device (the reference loops over
the timestamps with a boolean-index + shape read each, trajectory_filler.py:46).
"""
import torch

from . import lietorch_shim as lietorch
from .factor_graph import FactorGraph
from .frontend import keyframe_count, set_keyframe_count
from .lietorch_shim import SE3

BATCH = 16                 # frames refined together (trajectory_filler.py:99)
REFINE_UPDATES = 6         # motion-only updates per batch (:71)
IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)


class PoseTrajectoryFiller:
    """fills in the poses of non-keyframe images"""

    def __init__(self, net, video, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.count = 0
        self.video = video
        self.device = device
        self.MEAN = torch.tensor(IMAGENET_MEAN, device=device)[:, None, None]
        self.STDV = torch.tensor(IMAGENET_STD, device=device)[:, None, None]

    # ---- pieces of one batch ---------------------------------------------------------------------------------------
    def _bracket(self, tt, ts):
        """(t0, t1): last keyframe at or before each timestamp (-1 -> wraps to the newest, as the reference's
        `ts[ts <= t].shape[0] - 1` does) and its successor, clamped at the newest keyframe"""
        before = (ts.unsqueeze(0) <= tt.unsqueeze(1)).sum(dim=1) - 1
        after = torch.where(before < ts.shape[0] - 1, before + 1, before)
        return before, after

    def _interpolate(self, tt, ts, Ps, t0, t1):
        """constant-velocity pose between the bracketing keyframes: exp(log(P1 P0^-1) (t - t0)/(dt + 1e-3)) P0"""
        span = ts[t1] - ts[t0] + 1e-3
        twist = (Ps[t1] * Ps[t0].inv()).log() / span.unsqueeze(-1)
        return SE3.exp(twist * (tt - ts[t0]).unsqueeze(-1)) * Ps[t0]

    def _features(self, frames):
        """feature maps only -- no context network needed for motion-only refinement.  Normalises in place."""
        frames = frames.sub_(self.MEAN).div_(self.STDV)
        with torch.autocast("cuda", enabled=torch.device(self.device).type == "cuda"):
            return self.fnet(frames)

    def _refine(self, t0, t1, first, count):
        graph = FactorGraph(self.video, self.update, device=self.device)
        parked = torch.arange(first, first + count, device=self.device)
        for anchor in (t0, t1):                                   # one edge to each bracketing keyframe
            graph.add_factors(anchor.to(self.device), parked)
        for _ in range(REFINE_UPDATES):
            graph.update(first, first + count, motion_only=True)

    def _fill(self, timestamps, images, depths, intrinsics):
        v = self.video
        N, M = keyframe_count(v), len(timestamps)
        tt = torch.as_tensor(timestamps, device=self.device, dtype=v.timestamp.dtype)
        images = torch.stack(images, dim=0)                       # [M, b, 3, H, W]
        depths = None if depths is None else torch.stack(depths, dim=0)
        intrinsics = torch.stack(intrinsics, 0)
        frames = images.to(self.device)
        ts, Ps = v.timestamp[:N], SE3(v.poses[:N])
        t0, t1 = self._bracket(tt, ts)
        guess = self._interpolate(tt, ts, Ps, t0, t1)
        fmap = self._features(frames)
        set_keyframe_count(v, N + M)                              # park the batch behind the keyframes ...
        v[N:N + M] = (tt, images[:, 0], guess.data, 1, depths, intrinsics / 8.0, fmap)
        self._refine(t0, t1, N, M)
        refined = SE3(v.poses[N:N + M].clone())
        set_keyframe_count(v, N)                                  # ... and un-park it
        return [refined]

    @torch.no_grad()
    def __call__(self, image_stream):
        """image_stream yields (timestamp, image [b,3,H,W], depth | None, intrinsic [4], gt_pose); returns one SE3
        holding a world-to-camera pose per frame."""
        poses, batch = [], ([], [], [], [])

        def flush():
            stamps, imgs, deps, intr = batch
            poses.extend(self._fill(list(stamps), list(imgs), list(deps) if deps else None, list(intr)))
            for part in batch:
                part.clear()
        for timestamp, image, depth, intrinsic, _gt in image_stream:
            batch[0].append(timestamp)
            batch[1].append(image)
            if depth is not None:
                batch[2].append(depth)
            batch[3].append(intrinsic)
            if len(batch[0]) == BATCH:
                flush()
        if batch[0]:
            flush()
        return lietorch.cat(poses, dim=0)
