"""Update operator of the tracker (host side: PyTorch-ROCm module tree over the package's HIP kernels).

Mirrors the module tree of the reference's `DroidNet.update` (src/droid_net.py:70-140,
src/modules/gru.py:5-33, src/droid_net.py:34-67) so that a `droid.pth` state dict loads with
the same keys: update.{corr_encoder,flow_encoder,weight,delta,gru,agg}.  SURVEY.md 8(a5): these
convolutions were to stay ATen/MIOpen; since the end of round 1 the large 3x3 ones run on the package's own
implicit-GEMM kernel (gs_conv3x3_pp, see CONV3X3_IMPL below).

`torch_scatter.scatter_mean` (absent in this image) is replaced by an index_add segment mean.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class GradientClip(nn.Module):
    """Forward identity (the reference clips gradients in backward only, modules/clipping.py)."""

    def forward(self, x):
        return x


def segment_mean(x, index, num_segments):
    """scatter_mean(x, index, dim=0) for x [n,c,h,w] (src/droid_net.py:59 with the batch dim
    folded away), as one [segments x n] @ [n x c*h*w] GEMM with fp32 accumulation.  An
    index_add_ formulation costs 2.2 ms of fp16 atomics at E=75, 60x80 on MI355X; the GEMM reads
    x once (~30 us).  Keeps x's memory format (channels_last stays channels_last)."""
    n = x.shape[0]
    onehot = torch.zeros(num_segments, n, dtype=x.dtype, device=x.device)
    onehot[index, torch.arange(n, device=x.device)] = 1
    cnt = onehot.sum(dim=1, keepdim=True).clamp(min=1)
    avg = (onehot / cnt)
    if x.is_contiguous(memory_format=torch.channels_last):
        flat = x.permute(0, 2, 3, 1).reshape(n, -1)                     # free view of NHWC memory
        out = (avg @ flat).view(num_segments, x.shape[2], x.shape[3], x.shape[1])
        return out.permute(0, 3, 1, 2)                                  # logical NCHW, NHWC strides
    return (avg @ x.reshape(n, -1)).view((num_segments,) + tuple(x.shape[1:]))


def build_segments(index):
    """CSR grouping of edges by source keyframe for GraphAgg: returns dict(uniq, ix, offsets, order, n).
    `index` is the edge list ii; uniq/ix are torch.unique(ii, sorted=True, return_inverse=True)
    (src/droid_net.py:57), offsets/order (int32) list each segment's edges."""
    uniq, ix = torch.unique(index, sorted=True, return_inverse=True)
    order = torch.argsort(ix, stable=True).to(torch.int32)
    counts = torch.bincount(ix, minlength=uniq.numel())
    offsets = torch.zeros(uniq.numel() + 1, dtype=torch.int32, device=index.device)
    offsets[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return {"uniq": uniq, "ix": ix, "offsets": offsets, "order": order.contiguous(), "n": int(uniq.numel())}


def segment_mean_hip(x, seg, in_channel=0, channels=None, in_bias=None, in_relu=False):
    """scatter_mean over source keyframes for NHWC fp16 x [E,C,h,w] through gs_segment_mean (one
    HBM-bound pass: reads x once, writes the means).  Optionally reads the channel slice
    [in_channel, in_channel+channels) and applies relu(x + in_bias) on the fly."""
    from . import _lib
    E, C, h, w = x.shape
    c = C - in_channel if channels is None else channels
    out = torch.empty((seg["n"], c, h, w), dtype=x.dtype, device=x.device, memory_format=torch.channels_last)
    rc = _lib.lib().gs_segment_mean(x.data_ptr() + 2 * in_channel, C, _lib.ptr(in_bias), int(in_relu),
                                    _lib.ptr(seg["offsets"]), _lib.ptr(seg["order"]), _lib.ptr(out),
                                    seg["n"], h * w, c, _lib.stream_ptr(x.device))
    _lib.check(rc, "segment_mean")
    return out


def pack_head_weight(weight):
    """MFMA A-fragments for gs_conv3x3_head from a [O,128,3,3] convolution weight: fp16 [8,64,8] with
    wpack[ks][l][e] = W[o][16 ks + 8 (l>>5) + e][ky][kx], (l & 31) = (3 ky + kx) O + o."""
    O = weight.shape[0]
    assert weight.shape[1:] == (128, 3, 3) and O in (1, 2)
    w = weight.detach().float()
    pack = torch.zeros(8, 64, 8, dtype=torch.float32, device=w.device)
    cols = w.permute(2, 3, 0, 1).reshape(9 * O, 128)                 # [(ky,kx,o), ch]
    full = torch.zeros(32, 128, device=w.device)
    full[:9 * O] = cols
    full = full.view(32, 8, 2, 8)                                     # [col, ks, half, e]
    pack = full.permute(1, 2, 0, 3).reshape(8, 64, 8)                 # lane = half*32 + col
    return pack.half().contiguous()


def pack_1x1_weight(weight):
    """MFMA A-fragments for gs_conv1x1 from a [N,K,1,1] weight: fp16 [N/32][KS][64][8]."""
    N, K = weight.shape[0], weight.shape[1]
    KS = 8 if K <= 128 else 13
    w = torch.zeros(N, KS * 16, dtype=torch.float32, device=weight.device)
    w[:, :K] = weight.detach().float().reshape(N, K)
    w = w.view(N // 32, 32, KS, 2, 8)                                  # [nb, col, ks, half, e]
    return w.permute(0, 2, 3, 1, 4).reshape(N // 32, KS, 64, 8).half().contiguous()


def conv1x1_bias_act(cache, conv, x, act):
    """act(conv1x1(x) + bias) for NHWC fp16 x in one HIP launch (gs_conv1x1)."""
    from . import _lib
    key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        hit = (key, pack_1x1_weight(conv.weight), conv.bias.detach().float().contiguous())
        cache[id(conv)] = hit
    n, K, h, w = x.shape
    N = conv.weight.shape[0]
    y = torch.empty((n, N, h, w), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    rc = _lib.lib().gs_conv1x1(_lib.ptr(x), K, K, _lib.ptr(hit[1]), _lib.ptr(hit[2]), _ACT[act], _lib.ptr(y), N, N,
                               n * h * w, _lib.stream_ptr(x.device))
    _lib.check(rc, "conv1x1")
    return y


def pack_conv7x7_c4_weight(weight):
    """[128, 4, 7, 7] -> gs_conv7x7_c4's fp16 A-fragment image [2][2][14][64][8] (include/goslam_hip.h): K index
    k = 32 ky + 4 kx + c with an all-zero 8th tap per kernel row."""
    O, C, kh, kw = weight.shape
    assert (O, C, kh, kw) == (128, 4, 7, 7)
    wk = torch.zeros(O, 7, 8, 4, dtype=torch.float16, device=weight.device)
    wk[:, :, :7, :] = weight.detach().half().permute(0, 2, 3, 1)               # o ky kx c
    wk = wk.reshape(2, 2, 32, 14, 2, 8)                                        # nh t row s kg e
    return wk.permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)               # nh t s (kg row) e


def conv7x7_c4_supported(conv, x):
    return (x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and x.shape[1] == 4 and x.shape[3] <= 1024
            and tuple(conv.weight.shape) == (128, 4, 7, 7) and conv.stride in (1, (1, 1))
            and conv.padding in (3, (3, 3)) and x.is_contiguous(memory_format=torch.channels_last))


def conv7x7_c4_bias_act(cache, conv, x, act, rt=0):
    """act(conv7x7(x) + bias) for a 4-channel NHWC fp16 x in one HIP launch (gs_conv7x7_c4); act: none / relu."""
    from . import _lib
    key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        hit = (key, pack_conv7x7_c4_weight(conv.weight), conv.bias.detach().float().contiguous())
        cache[id(conv)] = hit
    n, _, h, w = x.shape
    y = torch.empty((n, 128, h, w), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.lib().gs_conv7x7_c4(_lib.ptr(x), _lib.ptr(hit[1]), _lib.ptr(hit[2]), _lib.ptr(y), 128, n, h, w,
                                      int(act == "relu"), int(rt), _lib.stream_ptr(x.device))
    _lib.check(rc, "conv7x7_c4")
    return y


def conv3x3_head(x, conv, cache, epilogue="none", out_scale=1.0, in_channel=0, in_bias=None, in_relu=False):
    """epi(conv(relu?(x[:, in_channel:in_channel+128] + in_bias)) + bias) * out_scale -> fp32 [n,h,w,O]
    (values are the reference's fp16 results), one HIP launch (gs_conv3x3_head)."""
    from . import _lib
    key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
    hit = cache.get(id(conv))
    if hit is None or hit[0] != key:
        hit = (key, pack_head_weight(conv.weight), conv.bias.detach().float().contiguous())
        cache[id(conv)] = hit
    n, C, h, w = x.shape
    O = conv.weight.shape[0]
    out = torch.empty(n, h, w, O, dtype=torch.float32, device=x.device)
    epi = {"none": 0, "sigmoid": 1, "softplus": 2}[epilogue]
    rc = _lib.lib().gs_conv3x3_head(x.data_ptr() + 2 * in_channel, C, _lib.ptr(in_bias), int(in_relu),
                                    _lib.ptr(hit[1]), _lib.ptr(hit[2]), O, epi, float(out_scale), _lib.ptr(out),
                                    n, h, w, _lib.stream_ptr(x.device))
    _lib.check(rc, "conv3x3_head")
    return out


def cvx_upsample(data, mask):
    """Convex 8x upsampling (src/droid_net.py:9-23): data [b,h,w,d], mask [b,576,h,w]."""
    b, h, w, d = data.shape
    data = data.permute(0, 3, 1, 2)
    wts = torch.softmax(mask.view(b, 1, 9, 8, 8, h, w), dim=2)
    nb = F.unfold(data, kernel_size=3, padding=1).view(b, d, 9, 1, 1, h, w)
    up = (wts * nb).sum(dim=2)                       # [b,d,8,8,h,w]
    return up.permute(0, 4, 2, 5, 3, 1).reshape(b, 8 * h, 8 * w, d)


class _HalfWeights:
    """fp16 / NHWC copies of conv weights (+ fp32 biases) for the inference fast path, refreshed when
    the parameters change (load_state_dict, .to, optimiser steps bump `_version`)."""

    def __init__(self):
        self._cache = {}

    def get(self, conv):
        key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
        hit = self._cache.get(id(conv))
        if hit is None or hit[0] != key:
            w = conv.weight.detach().half().contiguous(memory_format=torch.channels_last)
            b = conv.bias.detach().float().contiguous()
            hit = (key, w, b)
            self._cache[id(conv)] = hit
        return hit[1], hit[2]


_ACT = {"none": 0, "relu": 1, "sigmoid": 2}


def bias_act(y, b, act, out=None, out_channel=0, in_channel=0, channels=None):
    """act(y[:, in_channel:in_channel+channels] + b) through gs_bias_act; in place by default, or into
    out[:, out_channel:...] (both NHWC fp16, any channel counts that are multiples of 8)."""
    from . import _lib
    n, cy, h, wd = y.shape
    c = cy - in_channel if channels is None else channels
    dst, ldy, off = (y, cy, in_channel) if out is None else (out, out.shape[1], out_channel)
    rc = _lib.lib().gs_bias_act(y.data_ptr() + 2 * in_channel, _lib.ptr(b), dst.data_ptr() + 2 * off, n * h * wd, c,
                                cy, ldy, _ACT[act], _lib.stream_ptr(y.device))
    _lib.check(rc, "bias_act")
    return dst


# The large 3x3 convolutions of the inference fast path run on the package's own implicit-GEMM MFMA kernel
# (gs_conv3x3_pp, csrc/conv3x3_pp.hip: every layer, every map size).  CONV3X3_IMPL = "miopen" routes them through the
# library instead -- the referee of the GPU tests (own kernel vs MIOpen on the same operands), not a product setting
# and not read from the environment.  History of the A/B that retired the other variants (round-1 16x16-tile kernel,
# row-stacked variant, four-wave schedule, persistent grid): DESIGN.md 3b.
CONV3X3_IMPL = "own"
# Epilogues fused into the convolution kernel: the ConvGRU gate arithmetic (gs_conv3x3_gru_zr2 / _q) and bias + ReLU
# (gs_conv3x3_bias_relu); False = convolution + separate gate / bias kernels (the tests' referee for the fusion).
GRU_FUSED_EPILOGUE = True
# flow_encoder[0] (7x7, 4 -> 128) through gs_conv7x7_c4 (bias + ReLU fused); False: MIOpen + bias_act pass (referee)
CONV7X7_OWN = True
# ConvGRU global context: w(net) + sigmoid + pooling in one kernel (gs_gru_glo_fused); False: gs_conv1x1 + gs_gru_glo
GRU_GLO_FUSED = True
# GraphAgg's upmask convolution (128 -> 576) fused with DepthVideo.upsample's convex upsampling (gs_upmask_upsample): the
# update operator returns a LazyUpmask; False: gs_conv1x1 writes the mask, gs_cvx_upsample reads it (the tests' referee)
FUSE_UPMASK_UPSAMPLE = True
# correlation lookup fused with corr_encoder[0] (gs_corr_lookup_enc) when the caller hands a corr.LazyLookup; False:
# gs_corr_lookup_pyramid + gs_conv1x1 (the tests' referee for the fusion)
FUSE_LOOKUP_ENCODER = True
_CONV3X3_PACKS = {}        # (data_ptr, version, shape, device, kc) -> (packed image, weight tensor kept alive)
_CONV3X3_PACKS_MAX = 32


def pack_conv3x3_weight(weight, kc=32):
    """[O, C, 3, 3] -> gs_conv3x3_pp's fp16 LDS images [O/BN][C/32][9][4][BN][8] (include/goslam_hip.h):
    wpack[nb][ck][3 ky + kx][kg][r][e] = W[BN nb + r][32 ck + 8 kg + e][ky][kx]; BN = 128 output channels per workgroup,
    or 64 when O is only a multiple of 64."""
    O, C, kh, kw = weight.shape
    bn = 128 if O % 128 == 0 else 64
    assert (kh, kw) == (3, 3) and O % bn == 0 and C % kc == 0 and kc == 32
    w = weight.detach().half().reshape(O // bn, bn, C // kc, kc // 8, 8, 3, 3)     # nb r ck kg e ky kx
    return w.permute(0, 2, 5, 6, 3, 1, 4).contiguous().reshape(-1)                # nb ck ky kx kg r e


def conv3x3_hip_supported(x, w):
    """n_out % 64 == 0, c_in % 32 == 0, NHWC fp16"""
    return (x.is_cuda and x.dtype == torch.float16 and x.dim() == 4 and tuple(w.shape[2:]) == (3, 3)
            and w.shape[0] % 64 == 0 and w.shape[1] % 32 == 0 and x.shape[1] == w.shape[1]
            and x.is_contiguous(memory_format=torch.channels_last))


def _use_own_conv3x3(x, w, stride, padding):
    if CONV3X3_IMPL == "miopen" or stride not in (1, (1, 1)) or padding not in (1, (1, 1)):
        return False
    return conv3x3_hip_supported(x, w)


def conv3x3_weight_image(w, kc):
    """cached gs_conv3x3 weight image of `w` (see conv3x3_hip)"""
    key = (w.data_ptr(), w._version, tuple(w.shape), w.device, kc)
    hit = _CONV3X3_PACKS.pop(key, None)
    if hit is None:
        hit = (pack_conv3x3_weight(w, kc), w)
        while len(_CONV3X3_PACKS) >= _CONV3X3_PACKS_MAX:
            _CONV3X3_PACKS.pop(next(iter(_CONV3X3_PACKS)))
    _CONV3X3_PACKS[key] = hit                                # re-inserted last: least recently used goes first
    return hit[0]


def conv3x3_pp_tile_width(w):
    """tile width of the ping-pong kernel, 16 or 8: least column padding (ties: 16)"""
    return min((16, 8), key=lambda tw: ((w + tw - 1) // tw * tw, tw != 16))


def conv3x3_hip(x, w, tw=None):
    """bias-free 3x3 / pad 1 convolution of an NHWC fp16 tensor through gs_conv3x3_pp; `w` is the [O,C,3,3] weight.
    Its packed image is cached per (storage address, version, shape); the entry keeps the weight tensor alive, so the
    address cannot be recycled for different values while the entry exists."""
    from . import _lib
    image = conv3x3_weight_image(w, 32)
    n, c, h, wd = x.shape
    O = w.shape[0]
    y = torch.empty((n, O, h, wd), dtype=torch.float16, device=x.device, memory_format=torch.channels_last)
    with torch.cuda.device(x.device):
        rc = _lib.lib().gs_conv3x3_pp(_lib.ptr(x), c, c, _lib.ptr(image), tw or conv3x3_pp_tile_width(wd),
                                      _lib.ptr(y), O, O, n, h, wd, 1, _lib.stream_ptr(x.device))
    _lib.check(rc, "conv3x3")
    return y


def conv_nobias(x, w, stride=1, padding=0):
    if _use_own_conv3x3(x, w, stride, padding):
        return conv3x3_hip(x, w)
    with torch.autocast("cuda", enabled=False):
        y = F.conv2d(x, w, None, stride=stride, padding=padding)
    if not y.is_contiguous(memory_format=torch.channels_last):
        y = y.contiguous(memory_format=torch.channels_last)
    return y


def conv_bias_act(cache, conv, x, act, out=None, out_channel=0):
    """act(conv(x) + bias) for NHWC fp16 x: bias-free convolution (gs_conv3x3 or MIOpen) + one fused HIP epilogue
    (PyTorch issues conv, add_(bias) and relu_ as three passes).  With `out` (an NHWC fp16 tensor
    with more channels) the result lands in out[:, out_channel:out_channel+C] -- no torch.cat."""
    w, b = cache.get(conv)
    if GRU_FUSED_EPILOGUE and act == "relu" and _use_own_conv3x3(x, w, conv.stride, conv.padding):
        from . import _lib
        n, c, h, wd = x.shape
        O = w.shape[0]
        if out is None:
            out, out_channel = torch.empty((n, O, h, wd), dtype=torch.float16, device=x.device,
                                           memory_format=torch.channels_last), 0
        with torch.cuda.device(x.device):
            rc = _lib.lib().gs_conv3x3_bias_relu(_lib.ptr(x), c, c, _lib.ptr(conv3x3_weight_image(w, 32)), _lib.ptr(b),
                                                 out.data_ptr() + 2 * out_channel, out.shape[1], O, n, h, wd,
                                                 _lib.stream_ptr(x.device))
        _lib.check(rc, "conv3x3_bias_relu")
        return out
    y = conv_nobias(x, w, conv.stride, conv.padding)
    return bias_act(y, b, act, out, out_channel)


def copy_channels(x, out, out_channel):
    """out[:, out_channel:out_channel+C] = x for NHWC fp16 tensors (strided 16-byte copies)."""
    bias_act(x, None, "none", out, out_channel)


class LazyUpmask:
    """GraphAgg's upsampling mask [batch, M, 576, ht, wd], not evaluated yet: DepthVideo.upsample asks for
    `upsample_into(disps, ix, disps_up)` (1x1 convolution + softmax + 3x3 weighted sum in one launch, the 138 MB mask
    never exists); indexing ([0]) keeps it lazy; any other use materialises the ordinary fp16 mask."""

    def __init__(self, module, x, shape):
        self.module, self.x, self.shape_ = module, x, shape
        self._value = None

    def __getitem__(self, index):
        if index == 0 and self.shape_[0] == 1:
            return self
        return self.materialize()[index]

    def materialize(self):
        if self._value is None:
            m = self.module
            self._value = conv1x1_bias_act(m._head_cache, m.agg.upmask[0], self.x, "none").view(*self.shape_)
        return self._value

    def upsample_into(self, disps, ix, disps_up):
        from . import _lib
        conv = self.module.agg.upmask[0]
        cache = self.module._head_cache
        key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
        hit = cache.get("upmask_plain")
        if hit is None or hit[0] != key:
            hit = (key, conv.weight.detach().reshape(576, 128).half().contiguous(), conv.bias.detach().float().contiguous())
            cache["upmask_plain"] = hit
        x = self.x
        m, c, h, w = x.shape
        assert c == 128 and x.is_contiguous(memory_format=torch.channels_last) and ix.numel() == m
        with torch.cuda.device(x.device):
            rc = _lib.lib().gs_upmask_upsample(_lib.ptr(x), 128, _lib.ptr(hit[1]), _lib.ptr(hit[2]), _lib.ptr(disps),
                                               _lib.ptr(ix), _lib.ptr(disps_up), m, h, w, _lib.stream_ptr(x.device))
        _lib.check(rc, "upmask_upsample")

    def __getattr__(self, name):            # only reached for attributes not defined above
        return getattr(self.materialize(), name)


class ConvGRU(nn.Module):
    """src/modules/gru.py:5-33: ConvGRU with a global-context gate."""

    def __init__(self, h_planes=128, i_planes=128):
        super().__init__()
        self.do_checkpoint = False
        self.fuse_gates = True          # HIP gate fusion on the inference path (fp16, NHWC, no grad)
        c = h_planes + i_planes
        self.convz = nn.Conv2d(c, h_planes, 3, padding=1)
        self.convr = nn.Conv2d(c, h_planes, 3, padding=1)
        self.convq = nn.Conv2d(c, h_planes, 3, padding=1)
        self.w = nn.Conv2d(h_planes, h_planes, 1)
        self.convz_glo = nn.Conv2d(h_planes, h_planes, 1)
        self.convr_glo = nn.Conv2d(h_planes, h_planes, 1)
        self.convq_glo = nn.Conv2d(h_planes, h_planes, 1)

    def _fusable(self, net, inputs):
        cl = torch.channels_last
        return (net.is_cuda and not torch.is_grad_enabled() and net.dtype == torch.float16 and net.shape[1] == 128
                and net.is_contiguous(memory_format=cl) and all(t.dtype == torch.float16 for t in inputs))

    def _half_weights(self):
        """fp16 NHWC copies of the conv weights (convz|convr fused to one 448->256 conv), cached."""
        mods = (self.convz, self.convr, self.convq, self.w, self.convz_glo, self.convr_glo, self.convq_glo)
        key = tuple(m.weight._version for m in mods) + tuple(m.bias._version for m in mods) + \
            (self.convz.weight.device, self.convz.weight.data_ptr())
        if getattr(self, "_hw_key", None) != key:
            cl = torch.channels_last
            wzr = torch.cat([self.convz.weight, self.convr.weight], 0).detach().half().contiguous(memory_format=cl)
            wq = self.convq.weight.detach().half().contiguous(memory_format=cl)
            bzr = torch.cat([self.convz.bias, self.convr.bias]).detach().float().contiguous()
            bq = self.convq.bias.detach().float().contiguous()
            ww = self.w.weight.detach().half().contiguous(memory_format=cl)
            bw = self.w.bias.detach().float().contiguous()
            glo = [m.weight.detach().half().reshape(128, 128).contiguous()
                   for m in (self.convz_glo, self.convr_glo, self.convq_glo)]
            glo += [m.bias.detach().float().contiguous() for m in (self.convz_glo, self.convr_glo, self.convq_glo)]
            # hoisted form: the z|r|q convolutions split into the part over the constant context
            # features inp (input channels 128:256 -> one 128->384 conv, run once per edge set) and the
            # part over [net | corr | flow] (320 input channels, run every update)
            keep = list(range(0, 128)) + list(range(256, self.convz.weight.shape[1]))
            w_all = torch.cat([self.convz.weight, self.convr.weight, self.convq.weight], 0).detach()
            self._hw_hoist = (w_all[:, 128:256].half().contiguous(memory_format=cl),
                              w_all[:256][:, keep].half().contiguous(memory_format=cl),
                              w_all[256:][:, keep].half().contiguous(memory_format=cl))
            self._ww_pack = pack_1x1_weight(self.w.weight)
            self._hw, self._hw_key = (wzr, wq, bzr, bq, ww, bw, glo), key
        return self._hw

    def _forward_fused(self, net, inputs):
        """Same mathematics as forward(); the 3x3 convolutions go through conv_nobias, everything between them is
        HIP (gs_gru_glo, gs_gru_gate_zr, gs_gru_gate_q) and ONE 448-channel cat."""
        hx = torch.cat([net, *inputs], dim=1)
        if not hx.is_contiguous(memory_format=torch.channels_last):
            hx = hx.contiguous(memory_format=torch.channels_last)
        return self.forward_hx(net, hx)

    def inp_gates(self, inp):
        """[n,384,h,w] fp16 NHWC: the z | r | q convolutions restricted to the context features (bias-free).
        By linearity conv(W, [net|inp|corr|flow]) = conv(W_inp, inp) + conv(W_rest, [net|corr|flow]); inp
        is constant while an edge lives, so this term is computed once per edge set instead of in
        every update (2 x 28.6 % of the GRU's convolution FLOPs)."""
        self._half_weights()
        return conv_nobias(inp, self._hw_hoist[0], padding=1)

    def global_context(self, net):
        """(gzr [b,256], gq [b,128]) fp32: the global-context terms of the three gates (modules/gru.py:22-27)"""
        from . import _lib
        b, c, h, w = net.shape
        hw = h * w
        wzr, wq, bzr, bq, ww, bw, gw = self._half_weights()
        L = _lib.lib()
        st = _lib.stream_ptr(net.device)
        dev = net.device
        gzr = torch.empty(b, 256, dtype=torch.float32, device=dev)
        gq = torch.empty(b, 128, dtype=torch.float32, device=dev)
        if GRU_GLO_FUSED:
            # w(net), sigmoid, * net and the pooling in one kernel: w_pre never reaches memory
            ws = torch.empty(L.gs_gru_glo_fused_workspace_bytes(b, hw), dtype=torch.uint8, device=dev)
            _lib.check(L.gs_gru_glo_fused(_lib.ptr(net), 128, _lib.ptr(self._ww_pack), _lib.ptr(bw), _lib.ptr(gw[0]),
                                          _lib.ptr(gw[1]), _lib.ptr(gw[2]), _lib.ptr(gw[3]), _lib.ptr(gw[4]),
                                          _lib.ptr(gw[5]), _lib.ptr(gzr), _lib.ptr(gq), b, hw, _lib.ptr(ws),
                                          ws.numel(), st), "gru_glo_fused")
        else:
            # gs_conv1x1 (own MFMA kernel; deterministic, one launch) instead of an MIOpen 1x1 convolution, whose
            # solver choice -- and with it the fp16 rounding of the global-context gate -- can change between calls
            w_pre = torch.empty_like(net)
            _lib.check(L.gs_conv1x1(_lib.ptr(net), 128, 128, _lib.ptr(self._ww_pack), None, 0, _lib.ptr(w_pre), 128,
                                    128, b * hw, st), "conv1x1(gru.w)")
            ws = torch.empty(L.gs_gru_glo_workspace_bytes(b), dtype=torch.uint8, device=dev)
            _lib.check(L.gs_gru_glo(_lib.ptr(w_pre), _lib.ptr(bw), _lib.ptr(net), _lib.ptr(gw[0]), _lib.ptr(gw[1]),
                                    _lib.ptr(gw[2]), _lib.ptr(gw[3]), _lib.ptr(gw[4]), _lib.ptr(gw[5]),
                                    _lib.ptr(gzr), _lib.ptr(gq), b, hw, _lib.ptr(ws), ws.numel(), st), "gru_glo")
        return gzr, gq

    def forward_hx(self, net, hx, inp_pre=None, glo=None):
        """GRU step given hx = [net | inputs] already laid out as one NHWC fp16 tensor.  hx[:, :128] is
        overwritten with r*net (the q-convolution's input).  With `inp_pre` (= inp_gates(inp)) hx is
        [net | corr | flow] and the context-feature term is added inside the gate kernels."""
        from . import _lib
        b, c, h, w = net.shape
        hw = h * w
        wzr, wq, bzr, bq, ww, bw, gw = self._half_weights()
        if inp_pre is not None:
            wzr, wq = self._hw_hoist[1], self._hw_hoist[2]
        L = _lib.lib()
        st = _lib.stream_ptr(net.device)
        dev = net.device
        with torch.autocast("cuda", enabled=False):
            gzr, gq = glo if glo is not None else self.global_context(net)
            cin = hx.shape[1]
            if GRU_FUSED_EPILOGUE and _use_own_conv3x3(hx, wzr, 1, 1) and h * w > 0:
                # gate arithmetic in the convolutions' epilogues: zr_pre / q_pre never reach HBM; hx stays intact
                z = torch.empty_like(net)
                rnet = torch.empty_like(net)
                out = torch.empty_like(net)
                # (both convolutions take [net | hx[:, 128:]] as two tensors: net is never copied into hx[:, :128])
                _lib.check(L.gs_conv3x3_gru_zr2(_lib.ptr(net), hx.data_ptr() + 2 * 128, cin, cin - 128,
                                                _lib.ptr(conv3x3_weight_image(wzr, 32)), _lib.ptr(bzr), _lib.ptr(gzr),
                                                _lib.ptr(inp_pre), _lib.ptr(z), _lib.ptr(rnet), b, h, w, st),
                           "conv3x3_gru_zr2")
                _lib.check(L.gs_conv3x3_gru_q(_lib.ptr(rnet), hx.data_ptr() + 2 * 128, cin, cin - 128,
                                              _lib.ptr(conv3x3_weight_image(wq, 32)), _lib.ptr(bq), _lib.ptr(gq),
                                              _lib.ptr(inp_pre), _lib.ptr(z), _lib.ptr(net), _lib.ptr(out), b, h, w,
                                              st), "conv3x3_gru_q")
                return out
            copy_channels(net, hx, 0)                         # the unfused kernels read net from hx[:, :128]
            zr_pre = conv_nobias(hx, wzr, padding=1)
            z = torch.empty_like(net)
            _lib.check(L.gs_gru_gate_zr(_lib.ptr(zr_pre), _lib.ptr(bzr), _lib.ptr(gzr), _lib.ptr(inp_pre), _lib.ptr(hx),
                                        _lib.ptr(z), b, hw, hx.shape[1], st), "gru_gate_zr")
            q_pre = conv_nobias(hx, wq, padding=1)
            out = torch.empty_like(net)
            _lib.check(L.gs_gru_gate_q(_lib.ptr(q_pre), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z),
                                       _lib.ptr(net), _lib.ptr(out), b, hw, st), "gru_gate_q")
        return out

    def forward(self, net, *inputs):
        if self.fuse_gates and self._fusable(net, inputs):
            return self._forward_fused(net, inputs)
        inp = torch.cat(inputs, dim=1)
        hx = torch.cat([net, inp], dim=1)
        b, c, h, w = net.shape
        glo = (torch.sigmoid(self.w(net)) * net).view(b, c, h * w).mean(-1).view(b, c, 1, 1)
        z = torch.sigmoid(self.convz(hx) + self.convz_glo(glo))
        r = torch.sigmoid(self.convr(hx) + self.convr_glo(glo))
        q = torch.tanh(self.convq(torch.cat([r * net, inp], dim=1)) + self.convq_glo(glo))
        return (1 - z) * net + z * q


class GraphAgg(nn.Module):
    """src/droid_net.py:34-67: per-source-keyframe aggregation -> damping eta + upsampling mask."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(128, 128, 3, padding=1)
        self.conv2 = nn.Conv2d(128, 128, 3, padding=1)
        self.relu = nn.ReLU(inplace=True)
        self.eta = nn.Sequential(nn.Conv2d(128, 1, 3, padding=1), GradientClip(), nn.Softplus())
        self.upmask = nn.Sequential(nn.Conv2d(128, 8 * 8 * 9, 1, padding=0))

    def forward(self, net, ii):
        batch, num, ch, ht, wd = net.shape
        x = self.relu(self.conv1(net.view(batch * num, ch, ht, wd)))
        uniq, ix = torch.unique(ii, sorted=True, return_inverse=True)
        assert batch == 1
        x = segment_mean(x, ix, uniq.numel())
        x = self.relu(self.conv2(x))
        eta = self.eta(x).view(batch, -1, ht, wd)
        upmask = self.upmask(x).view(batch, -1, 8 * 8 * 9, ht, wd)
        return 0.01 * eta, upmask


class UpdateModule(nn.Module):
    """src/droid_net.py:70-140."""

    def __init__(self):
        super().__init__()
        cor_planes = 4 * (2 * 3 + 1) ** 2
        self.corr_encoder = nn.Sequential(
            nn.Conv2d(cor_planes, 128, 1), nn.ReLU(inplace=True),
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True))
        self.flow_encoder = nn.Sequential(
            nn.Conv2d(4, 128, 7, padding=3), nn.ReLU(inplace=True),
            nn.Conv2d(128, 64, 3, padding=1), nn.ReLU(inplace=True))
        self.weight = nn.Sequential(
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(128, 2, 3, padding=1), GradientClip(), nn.Sigmoid())
        self.delta = nn.Sequential(
            nn.Conv2d(128, 128, 3, padding=1), nn.ReLU(inplace=True),
            nn.Conv2d(128, 2, 3, padding=1), GradientClip())
        self.gru = ConvGRU(128, 128 + 128 + 64)
        self.agg = GraphAgg()
        self.fuse_epilogues = True      # inference fast path: bias-free convs + fused HIP epilogues
        self._hw = _HalfWeights()
        self._head_cache = {}

    def drop_edge_caches(self):
        """Forget what is cached per edge set (the hoisted context-feature convolutions).  Happens by itself when `inp`
        is a different tensor, was written through torch (version counter) or was freed; a caller that writes the
        features through a RAW POINTER (a HIP launch on `inp.data_ptr()`, another process on shared memory) calls this."""
        self._inp_pre_cache = None

    invalidate_context = drop_edge_caches       # the explicit hook (DepthVideo.add_write_hook(update.invalidate_context))

    # hoisted context terms kept at most: an eighth of the device's memory, never more than this (least recently used go
    # first; >= 1 entry stays).  An entry lives only as long as the `inp` tensor it was computed from (weak reference).
    INP_CACHE_BYTES = 16 << 30
    INP_CACHE_ENTRIES = 64          # ... and at most this many edge sets (a 200-keyframe global BA walks 16 chunks)
    # debug mode (GOSLAM_CACHE_CHECK=1 or `update.cache_check = True`): every cache hit re-reads `inp` and compares a
    # checksum with the one taken when the term was computed -- one reduction + one host sync per update, so a write the
    # version counter cannot see (raw pointer, other process) raises instead of silently serving the old term
    cache_check = os.environ.get("GOSLAM_CACHE_CHECK", "0") == "1"

    @staticmethod
    def _checksum(inp):
        flat = inp.detach().reshape(-1)
        if flat.dtype in (torch.float16, torch.bfloat16):
            flat = flat.view(torch.int16)
        elif flat.dtype == torch.float32:
            flat = flat.view(torch.int32)
        w = (torch.arange(flat.numel(), device=flat.device, dtype=torch.int64) % 8191) + 1      # position-sensitive
        return int((flat.to(torch.int64) * w).sum())

    _device_caps = {}               # device index -> an eighth of its memory (asked ONCE: the driver query takes tens to
                                    # hundreds of microseconds and this runs on every edge-set change)

    def _cache_cap(self, device):
        cap = self.INP_CACHE_BYTES
        if device.type == "cuda":
            idx = device.index if device.index is not None else torch.cuda.current_device()
            dev_cap = UpdateModule._device_caps.get(idx)
            if dev_cap is None:
                try:
                    dev_cap = torch.cuda.mem_get_info(device)[1] // 8
                except RuntimeError:
                    dev_cap = cap
                UpdateModule._device_caps[idx] = dev_cap
            cap = min(cap, dev_cap)
        return cap

    def _edge_state(self, inp, n, ht, wd):
        """(hx, inp_pre): the GRU's per-update input buffer [net | corr | flow] (320 ch, NHWC fp16) and
        the hoisted z|r|q convolutions over `inp`, recomputed only when `inp` is a different tensor or
        was written to -- in the factor graph that is when edges are added or removed.  The terms are cached PER `inp`
        TENSOR (round 5): FactorGraph.update_lowmem walks the same 13-keyframe chunks in every one of its steps, each with
        its own cached context tensor, and with one slot the 128 -> 384 convolution (+ a layout copy) was redone for every
        chunk of every step -- 16 x 123 us of a 16.9 ms stress step.  69 MB per 75-edge chunk at 30 x 40 (1.1 GB for the
        200-keyframe graph), 276 MB for the S480 frontend window.

        Lifetime (round 6): an entry holds a WEAK reference to the tensor that owns the features' memory (`inp`, or the
        tensor `inp` is a view of) and disappears with it -- FactorGraph.add_factors / rm_factors build a new `self.inp`
        on every edge-set change, and with strong references the frontend pinned up to 64 dead edge sets (8 GB at
        30 x 40, the 16 GB cap at 60 x 80).  While the owner lives its address cannot be handed to another tensor, so a
        hit on (address, shape, strides, dtype) + the shared version counter is the same memory with the same content.
        The byte cap follows the device (an eighth of its memory), and an out-of-memory while computing a term empties
        the cache and tries once more."""
        import collections
        import weakref
        hx = getattr(self, "_hx", None)
        if hx is None or hx.shape[0] != n or hx.shape[2:] != (ht, wd) or hx.device != inp.device:
            hx = torch.empty((n, 320, ht, wd), dtype=torch.float16, device=inp.device,
                             memory_format=torch.channels_last)
            self._hx = hx
        cache = getattr(self, "_inp_pre_cache", None)
        if cache is None:
            cache = self._inp_pre_cache = collections.OrderedDict()
        self.gru._half_weights()                              # refreshes gru._hw_key if the weights changed
        wkey = self.gru._hw_key
        # keyed by the MEMORY the features live in (address, shape, strides, dtype) and validated by the tensor's version
        # counter, which views share: `self.inp[None]` is a new Python object on every call (MotionFilter.track, 35 us per
        # input frame of recomputation with an object-identity key).
        owner = inp._base if inp._base is not None else inp
        key = (inp.data_ptr(), tuple(inp.shape), tuple(inp.stride()), inp.dtype)
        ent = cache.get(key)
        if ent is not None and ent[0]() is owner and ent[1] == inp._version and ent[2] == wkey:
            if self.cache_check and ent[4] != self._checksum(inp):
                raise RuntimeError("UpdateModule: the context features `inp` were modified behind the version counter "
                                   "(raw-pointer write?) while their hoisted convolution was cached; call "
                                   "update.invalidate_context() after such writes")
            cache.move_to_end(key)
            return hx, ent[3]
        inp4 = inp.view(n, -1, ht, wd)
        if inp4.dtype != torch.float16 or not inp4.is_contiguous(memory_format=torch.channels_last):
            inp4 = inp4.half().contiguous(memory_format=torch.channels_last)
        try:
            inp_pre = self.gru.inp_gates(inp4)
        except torch.OutOfMemoryError:
            cache.clear()
            torch.cuda.empty_cache()
            inp_pre = self.gru.inp_gates(inp4)
        for k in [k for k, e in cache.items() if e[2] != wkey or e[0]() is None]:       # old weights / dead owners
            del cache[k]

        def _gone(_ref, key=key, cache_ref=weakref.ref(cache)):
            c = cache_ref()
            if c is not None:
                e = c.get(key)
                if e is not None and e[0] is _ref:
                    del c[key]
        cache[key] = (weakref.ref(owner, _gone), inp._version, wkey, inp_pre,
                      self._checksum(inp) if self.cache_check else None)
        nbytes = lambda e: sum(x.numel() * x.element_size()
                               for x in (e[3] if isinstance(e[3], (tuple, list)) else (e[3],)) if torch.is_tensor(x))
        total = sum(nbytes(e) for e in cache.values())
        cap = self._cache_cap(inp.device)
        while (total > cap or len(cache) > self.INP_CACHE_ENTRIES) and len(cache) > 1:
            _, e = cache.popitem(last=False)
            total -= nbytes(e)
        return hx, inp_pre

    def _head_weights(self):
        """delta[0] | weight[0] | agg.conv1 read the same tensor: one 128->384 convolution."""
        mods = (self.delta[0], self.weight[0], self.agg.conv1)
        key = tuple(m.weight._version for m in mods) + tuple(m.bias._version for m in mods) + \
            (mods[0].weight.device, mods[0].weight.data_ptr())
        if getattr(self, "_heads_key", None) != key:
            w = torch.cat([m.weight for m in mods], 0).detach().half().contiguous(memory_format=torch.channels_last)
            b = [m.bias.detach().float().contiguous() for m in mods]
            self._heads, self._heads_key = (w, b), key
        return self._heads

    def _corr_independent_part(self, net4, inp, f4, n, ht, wd):
        """everything of the fast path that does not need the correlation features: the GRU input buffer with net and the
        flow-encoder features in place, the hoisted context term, the GRU's global-context terms"""
        hx, inp_pre = self._edge_state(inp, n, ht, wd)
        if CONV7X7_OWN and conv7x7_c4_supported(self.flow_encoder[0], f4):
            f4 = conv7x7_c4_bias_act(self._head_cache, self.flow_encoder[0], f4, "relu")
        else:
            f4 = conv_bias_act(self._hw, self.flow_encoder[0], f4, "relu")
        conv_bias_act(self._hw, self.flow_encoder[2], f4, "relu", out=hx, out_channel=256)
        with torch.autocast("cuda", enabled=False):
            glo = self.gru.global_context(net4)
        return hx, inp_pre, glo

    def _forward_fast(self, net, inp, corr, flow, ii, jj, seg=None):
        """forward() on the inference path: bias-free convolutions (gs_conv3x3 / MIOpen), each followed by one HIP
        epilogue (bias + activation, written straight into the next consumer's buffer -- no torch.cat),
        the context-feature part of the GRU convolutions hoisted out of the update loop, the three
        128->128 head convolutions merged into one; same mathematics, fp16 NHWC throughout."""
        batch, num, ch, ht, wd = net.shape
        cl = torch.channels_last
        hwc = self._hw
        out_dim = (batch, num, -1, ht, wd)
        n = batch * num
        net4 = net.view(n, -1, ht, wd)
        from .corr import LazyLookup
        lazy = corr if isinstance(corr, LazyLookup) and FUSE_LOOKUP_ENCODER and corr.block.fused_encoder_supported() \
            else None
        if lazy is None:
            if isinstance(corr, LazyLookup):
                corr = corr.materialize()
            c4 = corr.view(n, -1, ht, wd).half().contiguous(memory_format=cl)
        if flow is None:
            flow = torch.zeros(batch, num, 4, ht, wd, device=net.device)
        f4 = flow.view(n, -1, ht, wd).half().contiguous(memory_format=cl)
        # (running this part on a side stream next to the correlation lookup -- a light, memory-bound kernel -- was
        # measured: bit-identical, 2 % slower per keyframe; the kernels do not share the CUs to any advantage)
        hx, inp_pre, glo = self._corr_independent_part(net4, inp, f4, n, ht, wd)
        if lazy is not None:        # lookup + corr_encoder[0] in one launch: the 196-channel features never reach HBM
            c4 = lazy.encoded(*self._corr_enc0_padded())
        else:
            c4 = conv1x1_bias_act(self._head_cache, self.corr_encoder[0], c4, "relu")
        conv_bias_act(hwc, self.corr_encoder[2], c4, "relu", out=hx, out_channel=128)
        net4 = self.gru.forward_hx(net4, hx, inp_pre, glo=glo)
        net = net4.view(*out_dim)
        hw_, hb = self._head_weights()
        heads = conv_nobias(net4, hw_ if ii is not None else hw_[:256], padding=1)
        hc = self._head_cache
        # delta[2] / weight[2] read the merged convolution's pre-activations directly (bias + ReLU on the fly)
        delta = conv3x3_head(heads, self.delta[2], hc, "none", in_channel=0, in_bias=hb[0], in_relu=True)
        weight = conv3x3_head(heads, self.weight[2], hc, "sigmoid", in_channel=128, in_bias=hb[1], in_relu=True)
        delta, weight = delta.view(batch, num, ht, wd, 2), weight.view(batch, num, ht, wd, 2)
        if ii is None:
            return net, delta, weight
        # GraphAgg (src/droid_net.py:49-67)
        agg = self.agg
        if seg is None:
            seg = build_segments(ii.to(net.device))
        x = segment_mean_hip(heads, seg, in_channel=256, channels=128, in_bias=hb[2], in_relu=True)
        x = conv_bias_act(hwc, agg.conv2, x, "relu")
        eta = conv3x3_head(x, agg.eta[0], hc, "softplus", out_scale=0.01).view(batch, -1, ht, wd)
        if FUSE_UPMASK_UPSAMPLE:    # deferred: DepthVideo.upsample evaluates it fused with the convex upsampling
            upmask = LazyUpmask(self, x, (batch, -1, 8 * 8 * 9, ht, wd))
        else:
            upmask = conv1x1_bias_act(hc, agg.upmask[0], x, "none").view(batch, -1, 8 * 8 * 9, ht, wd)
        return net, delta, weight, eta, upmask

    def _corr_enc0_padded(self):
        """(fp16 [128, 208] weight of corr_encoder[0] with zero-padded rows, fp32 bias) for gs_corr_lookup_enc; cached"""
        conv = self.corr_encoder[0]
        key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
        hit = self._head_cache.get("corr_enc0_pad")
        if hit is None or hit[0] != key:
            w = torch.zeros(128, 208, dtype=torch.float16, device=conv.weight.device)
            w[:, :196] = conv.weight.detach().reshape(128, 196).half()
            hit = (key, w.contiguous(), conv.bias.detach().float().contiguous())
            self._head_cache["corr_enc0_pad"] = hit
        return hit[1], hit[2]

    def _fast_ok(self, net, inp, corr):
        cl = torch.channels_last
        n4 = net.view(-1, *net.shape[2:])
        return (self.fuse_epilogues and net.is_cuda and not torch.is_grad_enabled() and net.dtype == torch.float16
                and net.shape[2] == 128 and n4.is_contiguous(memory_format=cl) and torch.is_autocast_enabled())

    def forward(self, net, inp, corr, flow=None, ii=None, jj=None, seg=None):
        """`seg` (optional, not in the reference): build_segments(ii) cached by the caller, which
        saves the per-call torch.unique (a sort + host sync).  `corr` may be a corr.LazyLookup."""
        batch, num, ch, ht, wd = net.shape
        if self._fast_ok(net, inp, corr):
            return self._forward_fast(net, inp, corr, flow, ii, jj, seg)
        if hasattr(corr, "materialize"):
            corr = corr.materialize()
        if flow is None:
            flow = torch.zeros(batch, num, 4, ht, wd, device=net.device)
        out_dim = (batch, num, -1, ht, wd)
        net = net.view(batch * num, -1, ht, wd)
        inp = inp.view(batch * num, -1, ht, wd)
        corr = self.corr_encoder(corr.view(batch * num, -1, ht, wd))
        flow = self.flow_encoder(flow.view(batch * num, -1, ht, wd))
        net = self.gru(net, inp, corr, flow)
        delta = self.delta(net).view(*out_dim).permute(0, 1, 3, 4, 2)[..., :2].contiguous()
        weight = self.weight(net).view(*out_dim).permute(0, 1, 3, 4, 2)[..., :2].contiguous()
        net = net.view(*out_dim)
        if ii is not None:
            eta, upmask = self.agg(net, ii.to(net.device))
            return net, delta, weight, eta, upmask
        return net, delta, weight


class DroidNet(nn.Module):
    """fnet (instance-norm features, 128 ch) + cnet (context: 128 hidden + 128 input ch) + the update operator
    (src/droid_net.py:143-148); sub-module names are the checkpoint's."""

    def __init__(self):
        super().__init__()
        from .extractor import BasicEncoder
        self.fnet = BasicEncoder(out_dim=128, norm_fn="instance")
        self.cnet = BasicEncoder(out_dim=256, norm_fn="none")
        self.update = UpdateModule()


def load_pretrained(net, state_dict):
    """Load a DROID-SLAM checkpoint into `net` the way src/slam.py:196-208 does: strip the DataParallel `module.`
    prefix and keep only the first 2 output channels of the weight / delta heads (trained with 3)."""
    sd = {k.replace("module.", ""): v for k, v in state_dict.items()}
    for head in ("weight", "delta"):
        for p in ("weight", "bias"):
            k = f"update.{head}.2.{p}"
            sd[k] = sd[k][:2]
    net.load_state_dict(sd)
    return net
