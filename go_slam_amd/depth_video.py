"""Keyframe state buffers + the three hot-path methods of the reference's DepthVideo
(src/depth_video.py:207-269): `reproject`, `distance`, `ba` keep their signatures and
semantics; the native callees are the HIP kernels behind `go_slam_amd.droid_backends`.

This host mirror holds only what the hot path touches (poses, disps, disps_sens, intrinsics,
fmaps/nets/inps, disps_up); process-sharing flags and the mapping hand-off of the reference's
class are orchestration (out of scope).
"""
import torch

from . import droid_backends
from .droid_net import cvx_upsample


class DepthVideo:
    def __init__(self, ht, wd, buffer=512, device="cuda:0", stereo=False):
        self.device = torch.device(device)
        self.ht, self.wd = ht, wd            # 1/8-resolution map size
        d = self.device
        c = 2 if stereo else 1
        self.counter = 0                     # int here; a multiprocessing.Value (`.value`) is accepted too
        self.stereo = stereo
        self.timestamp = torch.zeros(buffer, device=d, dtype=torch.float32)
        self.poses = torch.zeros(buffer, 7, device=d, dtype=torch.float32)
        self.poses[:, 6] = 1.0
        self.disps = torch.ones(buffer, ht, wd, device=d, dtype=torch.float32)
        self.disps_sens = torch.zeros(buffer, ht, wd, device=d, dtype=torch.float32)
        self.disps_up = torch.zeros(buffer, 8 * ht, 8 * wd, device=d, dtype=torch.float32)
        self.intrinsics = torch.zeros(buffer, 4, device=d, dtype=torch.float32)
        self.fmaps = torch.zeros(buffer, c, 128, ht, wd, device=d, dtype=torch.half)
        self.nets = torch.zeros(buffer, 128, ht, wd, device=d, dtype=torch.half)
        self.inps = torch.zeros(buffer, 128, ht, wd, device=d, dtype=torch.half)
        self.dirty = torch.zeros(buffer, device=d, dtype=torch.bool)

    @staticmethod
    def format_indices(ii, jj, device="cuda"):
        if not isinstance(ii, torch.Tensor):
            ii = torch.as_tensor(ii)
        if not isinstance(jj, torch.Tensor):
            jj = torch.as_tensor(jj)
        return (ii.to(device=device, dtype=torch.long).reshape(-1).contiguous(),
                jj.to(device=device, dtype=torch.long).reshape(-1).contiguous())

    def upsample(self, ix, mask):
        """disps_up[ix] = cvx_upsample(disps[ix], mask) (src/depth_video.py:194-196).  fp16 masks go
        through the fused HIP kernel (no gather / unfold / softmax / index_put passes)."""
        m = mask.reshape(-1, 576, self.ht, self.wd)
        if m.dtype == torch.float16 and m.is_cuda:
            cl = m.is_contiguous(memory_format=torch.channels_last)
            if not cl:
                m = m.contiguous()
            ix = ix.to(device=self.device, dtype=torch.long).contiguous()
            from . import _lib
            with torch.cuda.device(self.device):
                rc = _lib.lib().gs_cvx_upsample(_lib.ptr(self.disps), _lib.ptr(m), _lib.ptr(ix),
                                                _lib.ptr(self.disps_up), ix.numel(), self.ht, self.wd, int(cl),
                                                _lib.stream_ptr(self.device))
            _lib.check(rc, "DepthVideo.upsample")
            return
        up = cvx_upsample(self.disps[ix].unsqueeze(-1), mask)
        self.disps_up[ix] = up.squeeze(-1).float()

    def reproject(self, ii, jj):
        """project points from ii -> jj: coords [1,E,h,w,2], valid [1,E,h,w,1]."""
        ii, jj = DepthVideo.format_indices(ii, jj, self.device)
        return droid_backends.reproject(self.poses, self.disps, self.intrinsics, ii, jj)

    def distance(self, ii=None, jj=None, beta=0.3, bidirectional=True):
        return_matrix = False
        N = int(getattr(self.counter, "value", self.counter))
        if ii is None:
            return_matrix = True
            ii, jj = torch.meshgrid(torch.arange(N), torch.arange(N), indexing="ij")
        ii, jj = DepthVideo.format_indices(ii, jj, self.device)
        intr = self.intrinsics[0].contiguous()
        if bidirectional:
            poses = self.poses[:N].clone()
            d1 = droid_backends.frame_distance(poses, self.disps, intr, ii, jj, beta)
            d2 = droid_backends.frame_distance(poses, self.disps, intr, jj, ii, beta)
            d = 0.5 * (d1 + d2)
        else:
            d = droid_backends.frame_distance(self.poses, self.disps, intr, ii, jj, beta)
        return d.reshape(N, N) if return_matrix else d

    def ba(self, target, weight, eta, ii, jj, t0=1, t1=None, iters=2, lm=1e-4, ep=0.1,
           motion_only=False, ba_type=None):
        """dense bundle adjustment (src/depth_video.py:257-269)."""
        if t1 is None:
            t1 = max(int(ii.max()), int(jj.max())) + 1
        out = droid_backends.ba(self.poses, self.disps, self.intrinsics[0].contiguous(), self.disps_sens,
                                target, weight, eta, ii, jj, t0, t1, iters, lm, ep, motion_only)
        self.disps.clamp_(min=0.001)
        return out
