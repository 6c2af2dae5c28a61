"""Correlation operators of the tracker, host side (mirrors src/modules/corr.py).

`CorrBlock` keeps the reference interface -- CorrBlock(fmap1, fmap2)(coords), `.cat`,
`[index]` -- but the 4 per-level sampler launches + 4 zero-fills + the concat of
`CorrBlock.__call__` (corr.py:43-53) are one fused HIP launch writing the [.,196,h,w] tensor.
"""
import torch
import torch.nn.functional as F

from . import _lib, droid_backends


class CorrBlock:
    def __init__(self, fmap1, fmap2, num_levels=4, radius=3, channels_last=False):
        assert num_levels == 4 and radius == 3, "GO-SLAM uses 4 levels, radius 3"
        self.num_levels = num_levels
        self.radius = radius
        self.channels_last = channels_last      # memory format of the looked-up features
        self.map_size = tuple(fmap2.shape[-2:])
        # tile8: levels 0-1 in 128-byte tiles (1.6x fewer L2 lines per lookup window); private to the
        # fused build -> lookup pair, so only used when both ends are the HIP kernels
        f1 = fmap1.reshape(-1, *fmap1.shape[2:])
        self.layout = droid_backends.CORR_TILE8 if (channels_last and fmap1.is_cuda and fmap1.dtype == torch.float16
                                                    and droid_backends.corr_tile8_supported(f1)) \
            else droid_backends.CORR_ROWMAJOR
        self.corr_pyramid = CorrBlock.build_pyramid(fmap1, fmap2, num_levels, self.layout)

    _warned_fallback = False

    @staticmethod
    def corr(fmap1, fmap2):
        """All-pairs correlation (corr.py:67-76): [batch,num,dim,ht,wd] x2 -> [batch,num,ht,wd,ht,wd]."""
        batch, num, dim, ht, wd = fmap1.shape
        f1 = fmap1.reshape(batch * num, dim, ht * wd) / 4.0
        f2 = fmap2.reshape(batch * num, dim, ht * wd) / 4.0
        return torch.matmul(f1.transpose(1, 2), f2).view(batch, num, ht, wd, ht, wd)

    @staticmethod
    def build_pyramid(fmap1, fmap2, num_levels=4, layout=0):
        batch, num, dim, ht, wd = fmap1.shape
        f1 = fmap1.reshape(batch * num, dim, ht, wd)
        if fmap1.is_cuda and num_levels == 4 and droid_backends.corr_volume_supported(f1):
            # fused HIP path: MFMA GEMM + the three pools, volume written exactly once
            return droid_backends.corr_volume_pyramid(f1.contiguous(),
                                                      fmap2.reshape(batch * num, dim, ht, wd).contiguous(), layout)
        assert layout == droid_backends.CORR_ROWMAJOR
        # shapes the fused kernel does not cover (w % 4 != 0, w > 96, other dtypes): the
        # reference's own formulation, a library GEMM + avg_pool2d.  Said out loud once per process: a reader of a
        # profile should not have to discover a hipBLASLt kernel on this path.
        if fmap1.is_cuda and not CorrBlock._warned_fallback:
            import warnings
            CorrBlock._warned_fallback = True
            warnings.warn(f"CorrBlock: {ht}x{wd} {fmap1.dtype} feature maps are outside gs_corr_volume_pyramid's shapes "
                          "(fp16, width a multiple of 4 up to 96); building the volume with torch.matmul + avg_pool2d",
                          RuntimeWarning, stacklevel=2)
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
        out = droid_backends.corr_lookup_pyramid(self.corr_pyramid, c, self.radius, self.channels_last,
                                                 self.layout, self.map_size)
        return out.view(batch, num, -1, ht, wd)

    def lazy(self, coords):
        """A deferred lookup for the update operator's fast path: the operator asks it for the looked-up features
        already passed through corr_encoder[0] (`encoded`: one fused launch, the 196-channel tensor never exists in HBM);
        any other use materialises the ordinary lookup."""
        return LazyLookup(self, coords)

    def fused_encoder_supported(self):
        p0 = self.corr_pyramid[0]
        return p0.is_cuda and p0.dtype == torch.float16 and self.channels_last

    def lookup_encoded(self, coords, wpad, bias):
        """relu(conv1x1(lookup(coords)) + bias): [batch*num, 128, ht, wd] fp16 NHWC (gs_corr_lookup_enc)"""
        from . import _lib
        batch, num, ht, wd, _ = coords.shape
        c = coords.reshape(batch * num, ht, wd, 2).float().contiguous()
        n = batch * num
        h2, w2 = self.map_size
        y = torch.empty((n, 128, ht, wd), dtype=torch.float16, device=c.device, memory_format=torch.channels_last)
        pyr = self.corr_pyramid
        with torch.cuda.device(c.device):
            rc = _lib.lib().gs_corr_lookup_enc(_lib.ptr(pyr[0]), _lib.ptr(pyr[1]), _lib.ptr(pyr[2]), _lib.ptr(pyr[3]),
                                               _lib.ptr(c), _lib.ptr(wpad), _lib.ptr(bias), _lib.ptr(y), 128, n, ht, wd,
                                               h2, w2, int(self.layout), _lib.stream_ptr(c.device))
        _lib.check(rc, "corr_lookup_enc")
        return y

    def cat(self, other):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = torch.cat([self.corr_pyramid[i], other.corr_pyramid[i]], dim=0)
        return self

    def __getitem__(self, index):
        for i in range(self.num_levels):
            self.corr_pyramid[i] = self.corr_pyramid[i][index].contiguous()
        return self


class CorrPool:
    """The correlation volumes of a factor graph's edges, with CorrBlock's interface (`(coords)`, `.lazy`, `.cat`,
    `[index]`, `.corr_pyramid`) but NO volume ever moved: level l lives in a capacity buffer [capacity, h, w, plane_l], edge e
    owns slot `slot[e]` of it (int64 device tensor, the lookups' extra argument), freed slots go to a device-side free list.

    The reference (and CorrBlock) keep the edges' volumes contiguous: `self.corr = self.corr.cat(corr)` on every
    add_factors and `self.corr = self.corr[~mask]` on every rm_factors (src/factor_graph.py:118,150) re-copy every volume
    held -- 61 MB per edge at 60 x 80, 4.6 GB at the 75-edge cap.  Measured on the end-to-end sequence leg (round 6,
    profiles/r06_frontend_e2e_kernel_stats*.md): 2.2 ms of `cat` + ~1 ms of gathers per keyframe around 7 ms of update
    operator.  Here `append` has gs_corr_volume_pyramid_slots write the new edges' planes straight into free slots and
    removal edits the slot list only.  Values and lookups are those of CorrBlock (same kernels, one more index)."""

    def __init__(self, ht, wd, device, capacity=96):
        self.num_levels, self.radius = 4, 3
        self.channels_last = True
        self.map_size = (int(ht), int(wd))
        self.device = torch.device(device)
        self.layout = droid_backends.CORR_TILE8 if wd % 16 == 0 else droid_backends.CORR_ROWMAJOR
        self.capacity = 0
        self.store = None
        self.slot = torch.zeros(0, dtype=torch.long, device=self.device)
        self.free = torch.zeros(0, dtype=torch.long, device=self.device)
        self._reserve(int(capacity))

    @staticmethod
    def supported(fmap1, channels_last=True):
        """fp16 CUDA feature maps of a shape the fused build covers, looked up into channels-last features"""
        return bool(channels_last and fmap1.is_cuda and fmap1.dtype == torch.float16
                    and droid_backends.corr_volume_supported(fmap1.reshape(-1, *fmap1.shape[-3:])))

    def _level_shape(self, l):
        ht, wd = self.map_size
        if self.layout == droid_backends.CORR_TILE8 and l < 2:
            return (ht, wd, int(_lib.lib().gs_corr_level_elems(ht, wd, l, int(self.layout))))
        return (ht, wd, ht >> l, wd >> l)

    def _reserve(self, capacity):
        """grow the buffers to `capacity` slots; live volumes keep their slot numbers (the only copy this class ever makes)"""
        new = [torch.empty((capacity,) + self._level_shape(l), dtype=torch.float16, device=self.device) for l in range(4)]
        if self.store is not None and self.slot.numel():
            for l in range(4):
                new[l].index_copy_(0, self.slot, self.store[l].index_select(0, self.slot))
        self.free = torch.cat([self.free, torch.arange(self.capacity, capacity, dtype=torch.long, device=self.device)])
        self.store, self.capacity = new, capacity

    def __len__(self):
        return int(self.slot.numel())

    def _take(self, n):
        if n > self.free.numel():
            self._reserve(max(self.capacity * 3 // 2, len(self) + n + 16))
        new, self.free = self.free[:n].contiguous(), self.free[n:]
        return new

    def append(self, fmap1, fmap2):
        """volumes + pooled levels of the edges fmap1[0, e] -> fmap2[0, e] ([1, n, 128, h, w] fp16), built in place"""
        f1 = fmap1.reshape(-1, *fmap1.shape[-3:]).contiguous()
        f2 = fmap2.reshape(-1, *fmap2.shape[-3:]).contiguous()
        n, dim, h, w = f1.shape
        assert (h, w) == self.map_size and f1.shape == f2.shape and f1.dtype == torch.float16
        if n == 0:
            return self
        new = self._take(n)
        L = _lib.lib()
        ws = droid_backends._workspace(self.device, L.gs_corr_volume_workspace_bytes(n, dim, h, w) + 256)
        with torch.cuda.device(self.device):
            rc = L.gs_corr_volume_pyramid_slots(_lib.ptr(f1), _lib.ptr(f2), *[_lib.ptr(v) for v in self.store], _lib.ptr(new),
                                                n, dim, h, w, int(self.layout), _lib.ptr(ws), ws.numel(),
                                                _lib.stream_ptr(self.device))
        _lib.check(rc, "CorrPool.append")
        self.slot = torch.cat([self.slot, new])
        return self

    def cat(self, other):
        """the reference's call form: `other` = an already built block of the same layout; its planes are copied into free slots"""
        pyr = other.corr_pyramid
        assert getattr(other, "layout", droid_backends.CORR_ROWMAJOR) == self.layout and tuple(other.map_size) == self.map_size
        new = self._take(int(pyr[0].shape[0]))
        for l in range(4):
            self.store[l].index_copy_(0, new, pyr[l].reshape((-1,) + self._level_shape(l)))
        self.slot = torch.cat([self.slot, new])
        return self

    def __getitem__(self, index):
        """keep the edges `index` selects (bool mask or indices), as CorrBlock[index]: slots change owner, no plane moves"""
        index = torch.as_tensor(index, device=self.device)
        if index.dtype == torch.bool:
            freed, self.slot = self.slot[~index], self.slot[index]
        else:
            index = index.long().reshape(-1)
            gone = torch.ones(len(self), dtype=torch.bool, device=self.device)
            gone[index] = False
            freed, self.slot = self.slot[gone], self.slot[index]
        self.free = torch.cat([freed, self.free])
        return self

    @property
    def corr_pyramid(self):
        """the edges' planes as CorrBlock holds them: a COPY, [E, h, w, ...] per level (tests, the materialised fallbacks)"""
        return [v.index_select(0, self.slot) for v in self.store]

    def fused_encoder_supported(self):
        return True

    def _coords(self, coords):
        batch, num, ht, wd, _ = coords.shape
        assert batch * num == len(self), (batch * num, len(self))
        return coords.reshape(batch * num, ht, wd, 2).float().contiguous(), batch, num, ht, wd

    def __call__(self, coords):
        c, batch, num, ht, wd = self._coords(coords)
        h2, w2 = self.map_size
        out = torch.empty((batch * num, 196, ht, wd), dtype=torch.float16, device=c.device, memory_format=torch.channels_last)
        s = self.store
        with torch.cuda.device(c.device):
            rc = _lib.lib().gs_corr_lookup_pyramid_slots(_lib.ptr(s[0]), _lib.ptr(s[1]), _lib.ptr(s[2]), _lib.ptr(s[3]),
                                                         _lib.ptr(self.slot), _lib.ptr(c), _lib.ptr(out), batch * num, ht, wd,
                                                         h2, w2, 3, droid_backends._DT[torch.float16], 1, int(self.layout),
                                                         _lib.stream_ptr(c.device))
        _lib.check(rc, "CorrPool lookup")
        return out.view(batch, num, -1, ht, wd)

    def lazy(self, coords):
        return LazyLookup(self, coords)

    def lookup_encoded(self, coords, wpad, bias):
        """relu(conv1x1(lookup(coords)) + bias): [batch*num, 128, ht, wd] fp16 NHWC (gs_corr_lookup_enc_slots)"""
        c, batch, num, ht, wd = self._coords(coords)
        n = batch * num
        h2, w2 = self.map_size
        y = torch.empty((n, 128, ht, wd), dtype=torch.float16, device=c.device, memory_format=torch.channels_last)
        s = self.store
        with torch.cuda.device(c.device):
            rc = _lib.lib().gs_corr_lookup_enc_slots(_lib.ptr(s[0]), _lib.ptr(s[1]), _lib.ptr(s[2]), _lib.ptr(s[3]),
                                                     _lib.ptr(self.slot), _lib.ptr(c), _lib.ptr(wpad), _lib.ptr(bias),
                                                     _lib.ptr(y), 128, n, ht, wd, h2, w2, int(self.layout),
                                                     _lib.stream_ptr(c.device))
        _lib.check(rc, "corr_lookup_enc")
        return y


class LazyLookup:
    """CorrBlock(coords), not evaluated yet.  `encoded(wpad, bias)` = the lookup fused with corr_encoder[0]; everything
    else (`.view`, `.float()`, tensor attributes ...) goes to the materialised [batch, num, 196, ht, wd] tensor."""

    def __init__(self, block, coords):
        self.block, self.coords = block, coords
        self._value = None

    def materialize(self):
        if self._value is None:
            self._value = self.block(self.coords)
        return self._value

    def encoded(self, wpad, bias):
        return self.block.lookup_encoded(self.coords, wpad, bias)

    def __getattr__(self, name):            # only reached for attributes not defined above
        return getattr(self.materialize(), name)


class AltCorrBlock:
    """Low-memory correlation (src/modules/corr.py:95-145): no [h,w,h,w] volumes; the 7x7 windows
    are correlated on the fly from a channels-last feature pyramid.  Same interface:
    AltCorrBlock(fmaps)(coords, ii, jj).  The pyramid stays fp16 (the reference casts it to fp32
    per call, corr.py:125; the kernel accumulates fp16 products in fp32, which is exact for them)."""

    def __init__(self, fmaps, num_levels=4, radius=3):
        self.num_levels = num_levels
        self.radius = radius
        B, N, C, H, W = fmaps.shape
        f = fmaps.reshape(B * N, C, H, W) / 4.0
        self.pyramid = []
        for i in range(num_levels):
            self.pyramid.append(f.permute(0, 2, 3, 1).contiguous().view(B, N, H // 2 ** i, W // 2 ** i, C))
            if i + 1 < num_levels:
                f = F.avg_pool2d(f, kernel_size=2, stride=2)

    def corr_fn(self, coords, ii, jj):
        B, N, H, W, S, _ = coords.shape
        coords = coords.permute(0, 1, 4, 2, 3, 5)
        out = []
        for i in range(self.num_levels):
            f1 = self.pyramid[0][:, ii]
            f2 = self.pyramid[i][:, jj]
            ci = (coords / 2 ** i).reshape(B * N, S, H, W, 2).contiguous()
            f1 = f1.reshape((B * N,) + f1.shape[2:]).contiguous()
            f2 = f2.reshape((B * N,) + f2.shape[2:]).contiguous()
            corr, = droid_backends.altcorr_forward(f1, f2, ci.float(), self.radius)
            out.append(corr.float().view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2))
        return torch.cat(out, dim=2)

    def fused_supported(self, coords):
        """shapes gs_altcorr_pyramid covers: the reference's configuration (fp16 pyramid of 128 channels, 4 levels, radius
        3, one coordinate set per edge, batch 1) on maps of at least 8 x 8"""
        p0 = self.pyramid[0]
        return (p0.is_cuda and p0.dtype == torch.float16 and p0.shape[-1] == 128 and self.num_levels == 4
                and self.radius == 3 and p0.shape[0] == 1 and coords.shape[0] == 1 and coords.dim() == 5
                and min(p0.shape[2], p0.shape[3]) >= 8)

    def lookup_fused(self, coords, ii, jj):
        """All four levels of one chunk of edges in ONE launch (csrc/altcorr_pyramid.hip): coords f32 [1, E, H, W, 2], ii /
        jj int64 [E] -> the [1, E, 196, H, W] features as an fp16 tensor in channels-last memory order -- what the update
        operator's corr_encoder[0] reads, with no gather, scale, cast, permute or cat launch around the kernel."""
        _, E, H, W, _ = coords.shape
        dev = coords.device
        out = torch.empty(E, H, W, 196, dtype=torch.float16, device=dev)
        c = coords.detach().float().contiguous()
        ii = ii.to(device=dev, dtype=torch.int64).contiguous()
        jj = jj.to(device=dev, dtype=torch.int64).contiguous()
        p = self.pyramid
        with torch.cuda.device(dev):
            rc = _lib.lib().gs_altcorr_pyramid(_lib.ptr(p[0]), _lib.ptr(p[1]), _lib.ptr(p[2]), _lib.ptr(p[3]), _lib.ptr(c),
                                               _lib.ptr(ii), _lib.ptr(jj), _lib.ptr(out), E, H, W, 128, self.radius,
                                               _lib.stream_ptr(dev))
        _lib.check(rc, "AltCorrBlock.lookup_fused")
        return out.permute(0, 3, 1, 2)[None]

    def lookup(self, coords, ii, jj):
        """What FactorGraph.update_lowmem calls: the one-launch fp16 / channels-last result when the shapes allow it, else
        `__call__`."""
        if self.fused_supported(coords):
            return self.lookup_fused(coords, ii, jj)
        return self(coords, ii, jj)

    def __call__(self, coords, ii, jj):
        """The reference's contract: a contiguous fp32 [B, N, 196, H, W(, S)] tensor."""
        squeeze = False
        if coords.dim() == 5:
            coords = coords.unsqueeze(-2)
            squeeze = True
        corr = self.corr_fn(coords, ii, jj)
        if squeeze:
            corr = corr.squeeze(-1)
        return corr.contiguous()
