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


class MotionFilter:
    """filters incoming frames and extracts their features"""

    def __init__(self, net, video, thresh=2.5, device="cuda:0"):
        self.cnet, self.fnet, self.update = net.cnet, net.fnet, net.update
        self.video = video
        self.thresh = thresh
        self.device = device
        self.count = 0
        self.MEAN = torch.tensor([0.485, 0.456, 0.406], device=device)[:, None, None]
        self.STDV = torch.tensor([0.229, 0.224, 0.225], device=device)[:, None, None]
        self._coords0 = None

    def _autocast(self):
        return torch.autocast("cuda", enabled=torch.device(self.device).type == "cuda")

    def _context_encoder(self, image):
        """[1, b, 3, H, W] -> hidden state (tanh) and input features (relu), each [b, 128, H/8, W/8]"""
        net, inp = self.cnet(image).split([128, 128], dim=2)
        return net.tanh().squeeze(0), inp.relu().squeeze(0)

    def _feature_encoder(self, image):
        return self.fnet(image).squeeze(0)

    @torch.no_grad()
    def track(self, timestamp, image, depth=None, intrinsic=None, gt_pose=None):
        """main update operation - run on every frame of the video (src/motion_filter.py:41-90).
        image: [b, 3, H, W] in [0, 1], b = 1 (mono / rgbd) or 2 (stereo, left first)."""
        scale = 8.0
        with self._autocast():
            ht, wd = image.shape[-2] // 8, image.shape[-1] // 8
            # As in the reference (motion_filter.py:52-53) the normalisation is in place on the device copy: a host
            # `image` is stored in the video unnormalised, an `image` already on the device aliases `inputs` and is
            # stored normalised.
            inputs = image.unsqueeze(0).to(self.device)
            inputs = inputs.sub_(self.MEAN).div_(self.STDV)
            gmap = self._feature_encoder(inputs)                          # [b, 128, ht, wd]
            left = 0                                                      # only the left view keeps net / inp
            if keyframe_count(self.video) == 0:                           # the first frame is always a keyframe
                net, inp = self._context_encoder(inputs[:, [left]])
                self.net, self.inp, self.fmap = net, inp, gmap
                ident = torch.tensor(IDENTITY_POSE, device=self.device)
                self.video.append(timestamp, image[left], ident, 1.0, depth, intrinsic / scale, gmap, net[left],
                                  inp[left], gt_pose)
                return
            if self._coords0 is None or self._coords0.shape[2:4] != (ht, wd):
                self._coords0 = coords_grid(ht, wd, self.device)[None, None]
            corr = CorrBlock(self.fmap[None, [left]], gmap[None, [left]])(self._coords0)
            # approximate flow magnitude with one update iteration
            _, delta, weight = self.update(self.net[None], self.inp[None], corr)
            if float(delta.norm(dim=-1).mean()) > self.thresh:            # enough motion: new keyframe
                self.count = 0
                net, inp = self._context_encoder(inputs[:, [left]])
                self.net, self.inp, self.fmap = net, inp, gmap
                self.video.append(timestamp, image[left], None, None, depth, intrinsic / scale, gmap, net[left],
                                  inp[left], gt_pose)
            else:
                self.count += 1
