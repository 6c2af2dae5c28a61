"""Feature / context encoders run once per input frame by MotionFilter.track (mirrors
src/modules/extractor.py:4-126).  Parameter names follow the reference's modules
(`conv1`, `norm1`, `layer{1,2,3}.{0,1}.{conv1,conv2,norm1,norm2,norm3,downsample.{0,1}}`, `conv2`) so a pretrained
`droid.pth` loads with `load_state_dict(strict=True)`.

The convolutions stay on MIOpen; what this module adds for MI355X is the memory format: `forward`
keeps every activation NHWC (channels_last) when the input is a CUDA half / autocast tensor, which selects MIOpen's
NHWC fp16 MFMA kernels and lets InstanceNorm / ReLU stream contiguous channel vectors.
"""
import torch
import torch.nn as nn

DIM = 32


def _norm(kind, planes, groups=None):
    if kind == "group":
        return nn.GroupNorm(num_groups=groups if groups is not None else planes // 8, num_channels=planes)
    if kind == "batch":
        return nn.BatchNorm2d(planes)
    if kind == "instance":
        return nn.InstanceNorm2d(planes)
    if kind == "none":
        return nn.Sequential()
    raise TypeError(kind)


class ResidualBlock(nn.Module):
    """two 3x3 convs + skip; the skip is a strided 1x1 conv (+ norm) when the block downsamples."""

    def __init__(self, in_planes, planes, norm_fn="group", stride=1):
        super().__init__()
        self.conv1 = nn.Conv2d(in_planes, planes, 3, stride=stride, padding=1)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.norm1 = _norm(norm_fn, planes)
        self.norm2 = _norm(norm_fn, planes)
        self.downsample = None
        if stride > 1:
            self.norm3 = _norm(norm_fn, planes)
            self.downsample = nn.Sequential(nn.Conv2d(in_planes, planes, 1, stride=stride), self.norm3)

    def forward(self, x):
        y = self.relu(self.norm1(self.conv1(x)))
        y = self.relu(self.norm2(self.conv2(y)))
        skip = x if self.downsample is None else self.downsample(x)
        return self.relu(skip + y)


class BasicEncoder(nn.Module):
    """[b, n, 3, H, W] -> [b, n, out_dim, H/8, W/8]: 7x7/2 stem, three pairs of residual blocks (32, 64/2, 128/2),
    1x1 projection."""

    def __init__(self, out_dim, norm_fn="batch"):
        super().__init__()
        self.out_dim, self.norm_fn = out_dim, norm_fn
        self.norm1 = _norm(norm_fn, DIM, groups=8)
        self.conv1 = nn.Conv2d(3, DIM, 7, stride=2, padding=3)
        self.relu1 = nn.ReLU(inplace=True)
        widths, planes = ((DIM, 1), (2 * DIM, 2), (4 * DIM, 2)), DIM
        for k, (dim, stride) in enumerate(widths, start=1):
            setattr(self, f"layer{k}", nn.Sequential(ResidualBlock(planes, dim, norm_fn, stride),
                                                     ResidualBlock(dim, dim, norm_fn, 1)))
            planes = dim
        self.conv2 = nn.Conv2d(4 * DIM, out_dim, 1)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
            elif isinstance(m, (nn.BatchNorm2d, nn.InstanceNorm2d, nn.GroupNorm)):
                if m.weight is not None:
                    nn.init.constant_(m.weight, 1)
                if m.bias is not None:
                    nn.init.constant_(m.bias, 0)

    def forward(self, x):
        b, n = x.shape[:2]
        x = x.reshape(b * n, *x.shape[2:])
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        x = self.conv2(x)
        return x.view(b, n, *x.shape[1:])
