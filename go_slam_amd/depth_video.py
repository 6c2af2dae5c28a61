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
    def __init__(self, ht, wd, buffer=512, device="cuda:0", stereo=False, full_res=False):
        """`full_res`: also keep the per-keyframe full-resolution buffers of the reference (images, depths_gt,
        poses_gt; src/depth_video.py:41-49) that MotionFilter.track fills -- 8h x 8w, allocated only on request."""
        self.device = torch.device(device)
        self.ht, self.wd = ht, wd            # 1/8-resolution map size
        d = self.device
        c = 2 if stereo else 1
        self.counter = 0                     # int here; a multiprocessing.Value (`.value`) is accepted too
        self.stereo = stereo
        H, W = 8 * ht, 8 * wd
        f32, f16 = torch.float32, torch.half
        # name -> (per-keyframe shape, dtype, fill): the 1/8-resolution state the hot path reads and writes ...
        table = {"timestamp": ((), f32, 0), "poses": ((7,), f32, 0), "disps": ((ht, wd), f32, 1),
                 "disps_sens": ((ht, wd), f32, 0), "disps_up": ((H, W), f32, 0), "intrinsics": ((4,), f32, 0),
                 "fmaps": ((c, 128, ht, wd), f16, 0), "nets": ((128, ht, wd), f16, 0), "inps": ((128, ht, wd), f16, 0),
                 "dirty": ((), torch.bool, 0)}
        if full_res:   # ... and the full-resolution / tracker -> mapper hand-off buffers (src/depth_video.py:41-69)
            table.update({"images": ((3, H, W), f32, 0), "depths_gt": ((H, W), f32, 0), "poses_gt": ((4, 4), f32, 0),
                          "poses_filtered": ((7,), f32, 0), "disps_filtered": ((H, W), f32, 0),
                          "mask_filtered": ((H, W), f32, 0), "update_priority": ((), f32, 0)})
        for name, (shape, dtype, fill) in table.items():
            setattr(self, name, torch.full((buffer,) + shape, fill, device=d, dtype=dtype))
        self.poses[:, 6] = 1.0               # identity: t = 0, q = (0, 0, 0, 1)
        if full_res:
            self.poses_gt[:] = torch.eye(4, device=d)
            self.poses_filtered[:, 6] = 1.0
            self.scale_factor = 8
            self.filtered_id = torch.tensor([-1], dtype=torch.int32, device=d)      # written by MultiviewFilter
            self.bound = torch.zeros(1, 3, 2, device=d, dtype=f32)
            self.pose_compensate = torch.tensor([[0, 0, 0, 0, 0, 0, 1.0]], device=d)

    @classmethod
    def from_config(cls, cfg, args):
        """the reference's constructor form DepthVideo(cfg, args) (src/depth_video.py:13-36)."""
        return cls(cfg["cam"]["H_out"] // 8, cfg["cam"]["W_out"] // 8, buffer=cfg["tracking"]["buffer"],
                   device=args.device, stereo=(cfg["mode"] == "stereo"), full_res=True)

    def get_lock(self):
        """single-process host mirror: nothing to lock (the reference guards IPC-shared buffers)."""
        import contextlib
        return contextlib.nullcontext()

    def _count(self):
        return int(getattr(self.counter, "value", self.counter))

    def _set_count(self, n):
        if hasattr(self.counter, "value"):
            self.counter.value = int(n)
        else:
            self.counter = int(n)

    def __setitem__(self, index, item):
        """item = (timestamp, image, pose | None, disp | None, depth | None, intrinsics | None[, fmap, net, inp,
        gt_pose]) -- src/depth_video.py:80-121.  A sensor depth is sub-sampled at the (3, 3) phase of every 8x8
        cell and becomes both the prior (disps_sens) and the initial disparity."""
        n = self._count()
        if isinstance(index, int) and index >= n:
            self._set_count(index + 1)
        elif torch.is_tensor(index) and int(index.max()) > n:
            self._set_count(int(index.max()) + 1)
        self.timestamp[index] = item[0]
        if hasattr(self, "images"):
            self.images[index] = item[1]
        pose, disp, depth = item[2], item[3], item[4]
        if pose is not None:
            self.poses[index] = pose
        if disp is not None:
            self.disps[index] = disp
        if depth is not None:                      # sensor depth: prior AND initial value (overrides `disp`)
            if hasattr(self, "depths_gt"):
                self.depths_gt[index] = depth
            sub = depth[..., 3::8, 3::8].to(self.device)
            self.disps_sens[index] = torch.where(sub > 0, 1.0 / sub, sub)
            self.disps[index] = self.disps_sens[index].clone()
        for slot, name in ((5, "intrinsics"), (6, "fmaps"), (7, "nets"), (8, "inps")):
            if len(item) > slot and (slot > 5 or item[slot] is not None):
                getattr(self, name)[index] = item[slot]
        if len(item) > 9 and item[9] is not None and hasattr(self, "poses_gt"):
            self.poses_gt[index] = item[9].to(self.poses_gt.device)

    def __getitem__(self, index):
        """(poses, disps, intrinsics, fmaps, nets, inps) at `index` (src/depth_video.py:127-143; the reference
        offsets POSITIVE int indices by the counter -- kept)."""
        if isinstance(index, int) and index > 0:
            index = self._count() + index
        return (self.poses[index], self.disps[index], self.intrinsics[index], self.fmaps[index], self.nets[index],
                self.inps[index])

    def append(self, *item):
        self[self._count()] = item

    def get_bound(self):
        return self.bound[0]

    def get_mapping_item(self, index, device="cuda:0", decay=0.1):
        """(image [H,W,3], depth [H,W], c2w [4,4], gt_c2w [4,4], mask [H,W]) of a filtered keyframe for the mapper
        (src/depth_video.py:153-177); each hand-out decays the keyframe's update priority."""
        from .lietorch_shim import SE3
        image = self.images[index].permute(1, 2, 0).contiguous().to(device)
        mask = self.mask_filtered[index].clone().to(device)
        depth = 1.0 / (self.disps_filtered[index].to(device) + 1e-7)
        w2c = SE3(self.poses_filtered[index].clone()).to(device)
        c2w = (SE3(self.pose_compensate[0].clone()).to(device) * w2c.inv()).matrix()      # origin alignment
        gt_c2w = self.poses_gt[index].clone().to(device)
        self.update_priority[index] *= decay
        return image, depth, c2w, gt_c2w, mask

    def normalize(self):
        """unit mean disparity over the keyframes so far; translations scale with it (src/depth_video.py:198-205)"""
        n = self._count()
        s = self.disps[:n].mean()
        self.disps[:n] /= s
        self.poses[:n, :3] *= s
        self.dirty[:n] = True

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
