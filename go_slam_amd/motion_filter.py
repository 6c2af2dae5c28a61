"""Per-frame keyframe selection + feature extraction (mirrors src/motion_filter.py:8-90): every input frame goes
through the feature encoder; a frame becomes a keyframe when one update-operator iteration against the previous
keyframe predicts a mean flow above `thresh`.

Hot-path pieces used here: CorrBlock (fused HIP volume build + one-launch lookup), the update operator, the
encoders (MIOpen, NHWC fp16).  One scalar crosses to the host per frame -- the keyframe decision itself.
"""
import torch

from .corr import CorrBlock
from .factor_graph import coords_grid
from .frontend import keyframe_count

IDENTITY_POSE = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0)        # lietorch.SE3.Identity(1).data: t = 0, q = (0,0,0,1)
LEFT = 0                                                   # stereo: only the left view keeps hidden / context state
DOWNSCALE = 8.0                                            # feature maps and stored intrinsics are at 1/8 resolution


class MotionFilter:
    """filters incoming frames and extracts their features"""

    def __init__(self, net, video, thresh=2.5, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.video = video
        self.thresh = thresh
        self.device = device
        self.count = 0                                       # frames skipped since the last keyframe
        self.MEAN = torch.tensor([0.485, 0.456, 0.406], device=device)[:, None, None]
        self.STDV = torch.tensor([0.229, 0.224, 0.225], device=device)[:, None, None]
        self._grid = None

    def _autocast(self):
        return torch.autocast("cuda", enabled=torch.device(self.device).type == "cuda")

    def _context(self, frames):
        """[1, b, 3, H, W] -> hidden state (tanh) and input features (relu), each [b, 128, H/8, W/8]"""
        hidden, ctx = self.cnet(frames).split([128, 128], dim=2)
        return hidden.tanh().squeeze(0), ctx.relu().squeeze(0)

    def _pixel_grid(self, ht, wd):
        if self._grid is None or tuple(self._grid.shape[2:4]) != (ht, wd):
            self._grid = coords_grid(ht, wd, self.device)[None, None]
        return self._grid

    def _flow_to_last_keyframe(self, gmap, ht, wd):
        """mean flow magnitude predicted by ONE update iteration from the last keyframe to this frame (host scalar)"""
        # (slices, not `[LEFT]` lists: a list index is a gather -- an index tensor, stride arithmetic on it and a copy
        # kernel, ~5 launches each on this per-frame path -- where the slice is a view)
        corr = CorrBlock(self.fmap[None, LEFT:LEFT + 1], gmap[None, LEFT:LEFT + 1])(self._pixel_grid(ht, wd))
        _, delta, _weight = self.update(self.net[None], self.inp[None], corr)
        return float(delta.norm(dim=-1).mean())

    def _make_keyframe(self, frames, gmap, timestamp, image, pose, disp, depth, intrinsic, gt_pose):
        hidden, ctx = self._context(frames[:, LEFT:LEFT + 1])
        self.net, self.inp, self.fmap = hidden, ctx, gmap
        self.video.append(timestamp, image[LEFT], pose, disp, depth, intrinsic / DOWNSCALE, gmap, hidden[LEFT],
                          ctx[LEFT], gt_pose)

    @torch.no_grad()
    def track(self, timestamp, image, depth=None, intrinsic=None, gt_pose=None):
        """main update operation - run on every frame of the video (src/motion_filter.py:41-90).
        image: [b, 3, H, W] in [0, 1], b = 1 (mono / rgbd) or 2 (stereo, left first)."""
        with self._autocast():
            ht, wd = image.shape[-2] // 8, image.shape[-1] // 8
            # As in the reference (motion_filter.py:52-53) the normalisation is in place on the device copy: a host
            # `image` is stored in the video unnormalised, an `image` already on the device aliases `frames` and is
            # stored normalised.
            frames = image.unsqueeze(0).to(self.device)
            frames = frames.sub_(self.MEAN).div_(self.STDV)
            gmap = self.fnet(frames).squeeze(0)                           # [b, 128, ht, wd]
            if keyframe_count(self.video) == 0:                           # the first frame is always a keyframe
                ident = torch.tensor(IDENTITY_POSE, device=self.device)
                self._make_keyframe(frames, gmap, timestamp, image, ident, 1.0, depth, intrinsic, gt_pose)
            elif self._flow_to_last_keyframe(gmap, ht, wd) > self.thresh:  # enough motion: new keyframe
                self.count = 0
                self._make_keyframe(frames, gmap, timestamp, image, None, None, depth, intrinsic, gt_pose)
            else:
                self.count += 1
