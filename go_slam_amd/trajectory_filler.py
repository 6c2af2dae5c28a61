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


class PoseTrajectoryFiller:
    """fills in the poses of non-keyframe images"""

    def __init__(self, net, video, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.count = 0
        self.video = video
        self.device = device
        self.MEAN = torch.tensor([0.485, 0.456, 0.406], device=device)[:, None, None]
        self.STDV = torch.tensor([0.229, 0.224, 0.225], device=device)[:, None, None]

    def _feature_encoder(self, image):
        with torch.autocast("cuda", enabled=torch.device(self.device).type == "cuda"):
            return self.fnet(image)

    def _fill(self, timestamps, images, depths, intrinsics):
        v = self.video
        tt = torch.as_tensor(timestamps, device=self.device, dtype=v.timestamp.dtype)
        images = torch.stack(images, dim=0)                       # [M, b, 3, H, W]
        depths = torch.stack(depths, dim=0) if depths is not None else None
        intrinsics = torch.stack(intrinsics, 0)
        inputs = images.to(self.device)
        N, M = keyframe_count(v), len(timestamps)
        ts = v.timestamp[:N]
        Ps = SE3(v.poses[:N])
        # last keyframe at or before each timestamp, and its successor (clamped at the end)
        t0 = (ts[None, :] <= tt[:, None]).sum(dim=1) - 1
        t1 = torch.where(t0 < N - 1, t0 + 1, t0)
        dt = ts[t1] - ts[t0] + 1e-3
        dP = Ps[t1] * Ps[t0].inv()
        vel = dP.log() / dt.unsqueeze(-1)
        Gs = SE3.exp(vel * (tt - ts[t0]).unsqueeze(-1)) * Ps[t0]
        inputs = inputs.sub_(self.MEAN).div_(self.STDV)
        fmap = self._feature_encoder(inputs)                      # no context features needed
        # park the non-keyframes behind the keyframes
        set_keyframe_count(v, N + M)
        v[N:N + M] = (tt, images[:, 0], Gs.data, 1, depths, intrinsics / 8.0, fmap)
        graph = FactorGraph(v, self.update, device=self.device)
        new = torch.arange(N, N + M, device=self.device)
        graph.add_factors(t0.to(self.device), new)
        graph.add_factors(t1.to(self.device), new)
        for _ in range(6):
            graph.update(N, N + M, motion_only=True)
        Gs = SE3(v.poses[N:N + M].clone())
        set_keyframe_count(v, N)
        return [Gs]

    @torch.no_grad()
    def __call__(self, image_stream):
        """image_stream yields (timestamp, image [b,3,H,W], depth | None, intrinsic [4], gt_pose); returns one SE3
        holding a world-to-camera pose per frame."""
        pose_list = []
        timestamps, images, depths, intrinsics = [], [], [], []

        def flush():
            nonlocal timestamps, images, depths, intrinsics
            pose_list.extend(self._fill(timestamps, images, depths if len(depths) > 0 else None, intrinsics))
            timestamps, images, depths, intrinsics = [], [], [], []
        for (timestamp, image, depth, intrinsic, gt_pose) in image_stream:
            timestamps.append(timestamp)
            images.append(image)
            if depth is not None:
                depths.append(depth)
            intrinsics.append(intrinsic)
            if len(timestamps) == 16:
                flush()
        if len(timestamps) > 0:
            flush()
        return lietorch.cat(pose_list, dim=0)
