"""Correlation operators of the tracker, host side (mirrors src/modules/corr.py).

`CorrBlock` keeps the reference interface -- CorrBlock(fmap1, fmap2)(coords), `.cat`,
`[index]` -- but the 4 per-level sampler launches + 4 zero-fills + the concat of
`CorrBlock.__call__` (corr.py:43-53) are one fused HIP launch writing the [.,196,h,w] tensor.
"""
import torch
import torch.nn.functional as F

from . import droid_backends


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, channels_last=False):
        assert num_levels == 4 and radius == 3, "GO-SLAM uses 4 levels, radius 3"
        self.num_levels = num_levels
        self.radius = radius
        self.channels_last = channels_last      # memory format of the looked-up features
        self.corr_pyramid = CorrBlock.build_pyramid(fmap1, fmap2, num_levels)

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation (corr.py:67-76): [batch,num,dim,ht,wd] x2 -> [batch,num,ht,wd,ht,wd]."""
        batch, num, dim, ht, wd = fmap1.shape
        f1 = fmap1.reshape(batch * num, dim, ht * wd) / 4.0
        f2 = fmap2.reshape(batch * num, dim, ht * wd) / 4.0
        return torch.matmul(f1.transpose(1, 2), f2).view(batch, num, ht, wd, ht, wd)

    @staticmethod
    def build_pyramid(fmap1, fmap2, num_levels=4):
        corr = CorrBlock.corr(fmap1, fmap2)
        batch, num, h1, w1, h2, w2 = corr.shape
        corr = corr.reshape(batch * num * h1 * w1, 1, h2, w2)
        pyramid = []
        for i in range(num_levels):
            pyramid.append(corr.view(batch * num, h1, w1, h2 // 2 ** i, w2 // 2 ** i))
            if i + 1 < num_levels:
                corr = F.avg_pool2d(corr, kernel_size=2, stride=2)
        return pyramid

    def __call__(self, coords):
        batch, num, ht, wd, _ = coords.shape
        c = coords.reshape(batch * num, ht, wd, 2).float().contiguous()
        out = droid_backends.corr_lookup_pyramid(self.corr_pyramid, c, self.radius, self.channels_last)
        return out.view(batch, num, -1, ht, wd)

    def cat(self, other):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = torch.cat([self.corr_pyramid[i], other.corr_pyramid[i]], dim=0)
        return self

    def __getitem__(self, index):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = self.corr_pyramid[i][index].contiguous()
        return self
