"""Feature / context encoders run once per input frame by MotionFilter.track (mirrors
src/modules/extractor.py:4-126).  Parameter names follow the reference's modules
(`conv1`, `norm1`, `layer{1,2,3}.{0,1}.{conv1,conv2,norm1,norm2,norm3,downsample.{0,1}}`, `conv2`) so a pretrained
`droid.pth` loads with `load_state_dict(strict=True)`.

What this module adds for MI355X:
  * the convolutions of the inference path on the package's own MFMA kernels (`gs_enc_conv`, csrc/enc_conv.hip: 7x7
    stride-2 stem, 3x3 at 32 / 64 / 128 channels with stride 1 / 2, strided 1x1 skips, 1x1 projection) -- no library
    convolution is left in `MotionFilter.track`; OWN_ENC_CONV = False (tests) routes them through MIOpen as the referee;
  * the memory format: every activation is NHWC (channels_last);
  * an inference path (`BasicEncoder._forward_fast`, used under no_grad on a GPU for norm_fn 'instance' / 'none'):
    fp16 NHWC weights cast ONCE (autocast re-casts weight and bias of all 11 convolutions on every call: 37 copy
    kernels per frame) and the elementwise tail of every convolution -- InstanceNorm + ReLU, and the block's
    `relu(skip + y)` -- in the three launches of `gs_norm_act` (csrc/instnorm.hip; one without the norm) instead of
    torch's 6-8 including the bias add: 200 launches per input frame become ~115, with the rounding points of the fp16 tensors the reference materialises.
"""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

DIM = 32
FAST_ENCODER = True         # module constant, not an environment switch: tests flip it to compare the two paths
OWN_ENC_CONV = True         # gs_enc_conv for the encoder's convolutions (False: MIOpen NHWC fp16, the tests' referee)
# the inference path as ONE hipGraph per (input shape, weight version): ~60 launches of 3-17 us per input frame replayed
# back to back instead of enqueued one by one (GOSLAM_ENCODER_GRAPHS=0: always eager)
ENCODER_GRAPHS = os.environ.get("GOSLAM_ENCODER_GRAPHS", "1") != "0"
_ENC_SHAPES = {(7, 4, 32, 2), (3, 32, 32, 1), (3, 32, 64, 2), (3, 64, 64, 1), (3, 64, 128, 2), (3, 128, 128, 1),
               (1, 32, 64, 2), (1, 64, 128, 2), (1, 128, 128, 1), (1, 128, 256, 1)}    # (k, c_in, c_out, stride) built


def pack_enc_conv_weight(weight):
    """[O, C, k, k] -> gs_enc_conv's MFMA A-fragments (include/goslam_hip.h).  k in {1, 3}: fp16 [k*k][C/16][O/32][64][8]
    with element [t][s][m][l][e] = W[32 m + (l & 31)][16 s + 8 (l >> 5) + e][t // k][t % k].  k = 7 (the stem, C = 3
    padded to the 4-channel RGB0 input): [7][2][64][8] with [dy][s][l][e] = W[l & 31][e % 4][dy][4 s + 2 (l >> 5) + e // 4],
    zero for channel 3 and for tap 7."""
    O, C, k, _ = weight.shape
    w = weight.detach().float()
    dev = w.device
    lane = torch.arange(64, device=dev)
    e = torch.arange(8, device=dev)
    if k == 7:
        assert O == 32 and C == 3
        wp = torch.zeros(7, 2, 64, 8, device=dev)
        for s_ in range(2):
            tap = 4 * s_ + 2 * (lane[:, None] >> 5) + e[None, :] // 4           # [64, 8]
            ch = (e[None, :] % 4).expand(64, 8)
            ok = (tap < 7) & (ch < 3)
            vals = w[(lane[:, None] & 31).expand(64, 8), ch.clamp(max=2), :, tap.clamp(max=6)]      # [64, 8, 7(dy)]
            wp[:, s_] = (vals * ok[..., None]).permute(2, 0, 1)
        return wp.half().contiguous()
    assert O % 32 == 0 and C % 16 == 0
    o_idx = (32 * torch.arange(O // 32, device=dev)[:, None, None] + (lane[None, :, None] & 31)).expand(O // 32, 64, 8)
    wp = torch.empty(k * k, C // 16, O // 32, 64, 8, device=dev)
    for s_ in range(C // 16):
        c_idx = (16 * s_ + 8 * (lane[None, :, None] >> 5) + e[None, None, :]).expand(O // 32, 64, 8)
        vals = w[o_idx, c_idx]                                                  # [O/32, 64, 8, k, k]
        wp[:, s_] = vals.reshape(O // 32, 64, 8, k * k).permute(3, 0, 1, 2)
    return wp.half().contiguous()
_WS = {}                    # (device index, stream) -> statistics workspace of gs_norm_act (two encoder calls on
                            # different streams -- tracking and backend threads -- must not share partial moments)


def _stats_workspace(device, nbytes):
    """per-(device, stream) grow-only workspace of gs_norm_act (chunk moments + mean / invstd)"""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _WS.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = _WS[key] = torch.empty(max(int(nbytes), 1 << 21), dtype=torch.uint8, device=device)
    return ws


def _norm_act(x, skip, instance, relu_in, relu_out, bias=None, stat_chunks=0):
    """in place on x (NHWC fp16 [n,c,h,w]): relu_out?(skip + relu_in?(instance_norm?(x + bias)))  -- gs_norm_act.
    stat_chunks > 0: the convolution that produced x (gs_enc_conv) already left its epilogue's chunk moments in this
    stream's workspace, so the statistics pass is skipped."""
    from . import _lib
    L = _lib.lib()
    n, c, h, w = x.shape
    ws = None
    if instance:
        nbytes = int(L.gs_norm_act_workspace_bytes_chunks(n, stat_chunks, c) if stat_chunks
                     else L.gs_norm_act_workspace_bytes(n, h * w, c))
        ws = _stats_workspace(x.device, nbytes)
    with torch.cuda.device(x.device):
        rc = L.gs_norm_act(_lib.ptr(x), _lib.ptr(bias), _lib.ptr(skip), _lib.ptr(x), n, h * w, c, int(instance), int(relu_in),
                           int(relu_out), 1e-5, _lib.ptr(ws), ws.numel() if ws is not None else 0, int(stat_chunks),
                           _lib.stream_ptr(x.device))
    _lib.check(rc, "norm_act")
    return x


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

    def _conv(self, conv, x, with_bias=True, stats=None):
        """conv(x) on fp16 NHWC operands cast / packed once per weight version (what autocast computes, minus the casts).
        with_bias=False returns (conv(x) without the bias, fp16 bias): gs_norm_act adds it on the fly (same fp16 rounding
        of conv + bias as the two fp16 tensors torch materialises).  stats=True (own kernels, InstanceNorm next): the
        convolution's epilogue also leaves the chunk moments of half(conv + bias) in this stream's norm workspace.  With `stats`
        given (True or False) a third value is returned: the `stat_chunks` to hand to _norm_act (0: none were written)."""
        cache = self.__dict__.setdefault("_w16", {})
        key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            hit = [key, conv.weight.detach().half().contiguous(memory_format=torch.channels_last),
                   conv.bias.detach().half().contiguous(), None]
            cache[id(conv)] = hit
        k, stride = conv.kernel_size[0], conv.stride[0]
        n, c, h, w = x.shape
        shape = (k, c, conv.out_channels, stride)
        if (OWN_ENC_CONV and shape in _ENC_SHAPES and x.dtype == torch.float16
                and x.is_contiguous(memory_format=torch.channels_last) and conv.padding[0] == k // 2):
            from . import _lib
            if hit[3] is None:
                hit[3] = pack_enc_conv_weight(conv.weight)
            ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
            y = torch.empty(n, conv.out_channels, ho, wo, dtype=torch.float16, device=x.device,
                            memory_format=torch.channels_last)
            L = _lib.lib()
            chunks, ws = 0, None
            if stats:
                chunks = int(L.gs_enc_conv_stat_chunks(ho, wo, conv.out_channels))
                ws = _stats_workspace(x.device, int(L.gs_norm_act_workspace_bytes_chunks(n, chunks, conv.out_channels)))
            with torch.cuda.device(x.device):
                rc = L.gs_enc_conv(_lib.ptr(x), c, c, _lib.ptr(hit[3]), _lib.ptr(hit[2] if with_bias else None),
                                   _lib.ptr(y), conv.out_channels, conv.out_channels, k, stride, n, h, w,
                                   _lib.ptr(hit[2] if stats else None), _lib.ptr(ws), _lib.stream_ptr(x.device))
            _lib.check(rc, "enc_conv")
            if stats is not None:
                return y, hit[2], chunks
            return y if with_bias else (y, hit[2])
        if c == 4 and conv.in_channels == 3:            # (the stem's RGB0 input on the library path)
            x = x[:, :3].contiguous(memory_format=torch.channels_last)
        if with_bias:
            return F.conv2d(x, hit[1], hit[2], conv.stride, conv.padding)
        if stats is not None:
            return F.conv2d(x, hit[1], None, conv.stride, conv.padding), hit[2], 0
        return F.conv2d(x, hit[1], None, conv.stride, conv.padding), hit[2]

    def _fast_ok(self, x):
        return (FAST_ENCODER and x.is_cuda and not torch.is_grad_enabled() and self.norm_fn in ("instance", "none")
                and (x.dtype == torch.float16 or torch.is_autocast_enabled()) and x.shape[-1] % 8 == 0
                and x.shape[-2] % 8 == 0)

    def _packs(self, conv):
        """(fp16 NHWC weight, fp16 bias, gs_enc_conv fragments) of a convolution, cached per weight version"""
        cache = self.__dict__.setdefault("_w16", {})
        key = (conv.weight._version, conv.bias._version, conv.weight.device, conv.weight.data_ptr())
        hit = cache.get(id(conv))
        if hit is None or hit[0] != key:
            hit = [key, conv.weight.detach().half().contiguous(memory_format=torch.channels_last),
                   conv.bias.detach().half().contiguous(), None]
            cache[id(conv)] = hit
        if hit[3] is None and OWN_ENC_CONV:
            hit[3] = pack_enc_conv_weight(conv.weight)
        return hit

    def _forward_fast(self, x):
        """The inference path.  One input frame is ~45 launches of 3-12 us each, so the HOST side decides the frame time:
        with the own kernels everywhere the stream handle, the device guard and the statistics workspace are taken once
        per call and the launches go straight to the C ABI with raw pointers (measured: ~4 us of Python per launch
        instead of ~10)."""
        inst = self.norm_fn == "instance"
        own = OWN_ENC_CONV
        with torch.autocast("cuda", enabled=False):
            n_, _, h_, w_ = x.shape                   # RGB0: the stem kernel reads 8-byte pixels
            dev = x.device
            x4 = torch.empty(n_, 4, h_, w_, dtype=torch.float16, device=dev, memory_format=torch.channels_last)
            x4[:, 3].zero_()
            x4[:, :3].copy_(x)
            if not own:
                return self._forward_fast_generic(x4, inst)
            from . import _lib
            L = _lib.lib()
            st = torch.cuda.current_stream(dev).cuda_stream
            # statistics workspace: the largest of the three resolutions decides (sizes from the library, once per shape)
            need = self.__dict__.setdefault("_ws_need", {}).get((n_, h_, w_))
            if need is None:
                need = 0
                for k_, c_ in ((2, 32), (4, 64), (8, 128)):
                    ho_, wo_ = -(-h_ // k_), -(-w_ // k_)
                    need = max(need, int(L.gs_norm_act_workspace_bytes_chunks(
                        n_, int(L.gs_enc_conv_stat_chunks(ho_, wo_, c_)), c_)))
                self._ws_need[(n_, h_, w_)] = need
            ws = _stats_workspace(dev, need)
            wsp, wsn = ws.data_ptr(), ws.numel()
            cl = torch.channels_last
            chunk_cache = self.__dict__.setdefault("_stat_chunks", {})
            # (convolution and normalisation are two launches sharing the stream's ONE statistics workspace: this path
            # is single-threaded per stream -- two host threads encoding on the same stream would need a workspace each)

            def conv(m, xin, stats):
                k, stride, co = m.kernel_size[0], m.stride[0], m.out_channels
                n, c, h, w = xin.shape
                hit = self._packs(m)
                ho, wo = (h + 2 * (k // 2) - k) // stride + 1, (w + 2 * (k // 2) - k) // stride + 1
                y = torch.empty(n, co, ho, wo, dtype=torch.float16, device=dev, memory_format=cl)
                b = hit[2].data_ptr()
                rc = L.gs_enc_conv(xin.data_ptr(), c, c, hit[3].data_ptr(), None, y.data_ptr(), co, co, k, stride, n, h, w,
                                   b if stats else None, wsp if stats else None, st)
                if rc:
                    _lib.check(rc, "enc_conv")
                if not stats:
                    return y, b, 0
                ch = chunk_cache.get((ho, wo, co))          # the kernel's own grid mapping, asked once per layer shape
                if ch is None:
                    ch = chunk_cache[(ho, wo, co)] = int(L.gs_enc_conv_stat_chunks(ho, wo, co))
                return y, b, ch

            def norm(t, skip, relu_in, relu_out, b, chunks):
                n, c, h, w = t.shape
                rc = L.gs_norm_act(t.data_ptr(), b, None if skip is None else skip.data_ptr(), t.data_ptr(), n, h * w, c,
                                   int(inst), int(relu_in), int(relu_out), 1e-5, wsp, wsn, chunks, st)
                if rc:
                    _lib.check(rc, "norm_act")
                return t

            with torch.cuda.device(dev):
                t, b, ch = conv(self.conv1, x4, inst)
                y = norm(t, None, True, False, b, ch)
                for layer in (self.layer1, self.layer2, self.layer3):
                    for blk in layer:
                        t, b, ch = conv(blk.conv1, y, inst)
                        t = norm(t, None, True, False, b, ch)
                        skip = y
                        if blk.downsample is not None:
                            # (before conv2: a convolution's epilogue statistics live in the stream's ONE norm workspace
                            # until the normalisation that follows it has consumed them)
                            skip, bs, chs = conv(blk.downsample[0], y, inst)
                            skip = norm(skip, None, False, False, bs, chs)
                        t, b, ch = conv(blk.conv2, t, inst)
                        y = norm(t, skip, True, True, b, ch)
            return self._conv(self.conv2, y)

    def _forward_fast_generic(self, x4, inst):
        """the same sequence through _conv / _norm_act (library convolutions: OWN_ENC_CONV = False, the tests' referee)"""
        t, b, ch = self._conv(self.conv1, x4, False, stats=inst)
        y = _norm_act(t, None, inst, True, False, b, ch)
        for layer in (self.layer1, self.layer2, self.layer3):
            for blk in layer:
                t, b, ch = self._conv(blk.conv1, y, False, stats=inst)
                t = _norm_act(t, None, inst, True, False, b, ch)
                skip = y
                if blk.downsample is not None:
                    skip, bs, chs = self._conv(blk.downsample[0], y, False, stats=inst)
                    skip = _norm_act(skip, None, inst, False, False, bs, chs)
                t, b, ch = self._conv(blk.conv2, t, False, stats=inst)
                y = _norm_act(t, skip, inst, True, True, b, ch)
        return self._conv(self.conv2, y)

    def _weights_key(self):
        ps = self.__dict__.get("_graph_params")
        if ps is None:
            ps = self.__dict__["_graph_params"] = list(self.parameters())
        return sum(p._version for p in ps), sum(p.data_ptr() for p in ps)

    def _forward_graphed(self, x):
        """_forward_fast(x) replayed from a hipGraph: captured per (input shape, dtype, device, weight versions and
        addresses) after two eager calls (workspaces, packed weights); the input is copied into the graph's static
        buffer and the result is returned as a fresh tensor (the static output is overwritten by the next replay).  A
        capture that fails (a library fallback that allocates with the driver, a foreign stream state) is recorded in
        `graph_error` and the shape stays eager."""
        graphs = self.__dict__.setdefault("_graphs", {})
        key = (tuple(x.shape), x.dtype, x.device, torch.is_autocast_enabled()) + self._weights_key()
        ent = graphs.get(key)
        if ent is None:
            if len(graphs) >= 4:                        # old weight versions / shapes: drop their graphs and pools
                graphs.clear()
            ent = graphs[key] = {"warm": 0, "graph": None, "failed": False}
        if ent["failed"]:
            return self._forward_fast(x)
        if ent["graph"] is None:
            if ent["warm"] < 2:
                ent["warm"] += 1
                return self._forward_fast(x)
            try:
                static_in = torch.empty_like(x, memory_format=torch.contiguous_format)
                static_in.copy_(x)
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    static_out = self._forward_fast(static_in)
                ent.update(graph=g, inp=static_in, out=static_out)
            except Exception as exc:                    # noqa: BLE001 -- whatever refused the capture: this shape stays eager
                ent["failed"] = True
                self.graph_error = repr(exc)[:300]
                torch.cuda.synchronize(x.device)
                return self._forward_fast(x)
        ent["inp"].copy_(x)
        ent["graph"].replay()
        return ent["out"].clone()

    def forward(self, x):
        b, n = x.shape[:2]
        x = x.reshape(b * n, *x.shape[2:])
        if self._fast_ok(x):
            if ENCODER_GRAPHS and OWN_ENC_CONV and not torch.cuda.is_current_stream_capturing():
                x = self._forward_graphed(x)
            else:
                x = self._forward_fast(x)
            return x.view(b, n, *x.shape[1:])
        if x.is_cuda:
            x = x.contiguous(memory_format=torch.channels_last)
        x = self.relu1(self.norm1(self.conv1(x)))
        x = self.layer3(self.layer2(self.layer1(x)))
        x = self.conv2(x)
        return x.view(b, n, *x.shape[1:])
