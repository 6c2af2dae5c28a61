"""Drop-in for the reference's `droid_backends` extension module
(reference: src/lib/droid.cpp:237-250 -- same nine names, argument order, dtypes, return
structure and in-place behaviour), implemented on the C-ABI HIP library.

    ba, frame_distance, projmap, depth_filter, iproj,
    corr_index_forward, corr_index_backward, altcorr_forward, altcorr_backward

Error behaviour mirrors `CHECK_CONTIGUOUS` (droid.cpp:84-85): a non-contiguous tensor raises
RuntimeError("<name> must be contiguous").  Unlike the reference (legacy default stream) the
kernels are enqueued on torch's *current* stream of the tensors' device.
"""
import os

import torch

from . import _lib

_DT = {torch.float16: 0, torch.float32: 1, torch.float64: 2}

_ws_cache = {}
_ba_status = {}            # device index -> int32[4] status of the last ba() enqueued there (still on the device)
# GOSLAM_BA_CHECK=1: read the status word back after every ba() (one host sync per call) and raise on a depth-row
# mismatch, as the reference's shape errors would; off by default -- ba_status() reads it on demand.
BA_CHECK = os.environ.get("GOSLAM_BA_CHECK", "0") == "1"


def _chk(name, t, dtype=None):
    if not isinstance(t, torch.Tensor):
        raise RuntimeError(f"{name} must be a tensor")
    if not t.is_cuda:
        raise RuntimeError(f"{name} must be a GPU tensor (go_slam_amd has no CPU path)")
    if not t.is_contiguous():
        raise RuntimeError(f"{name} must be contiguous")
    if dtype is not None and t.dtype != dtype:
        raise RuntimeError(f"{name} must have dtype {dtype}, got {t.dtype}")
    return t


def _workspace(device, nbytes):
    """Per-(device, stream) grow-only scratch buffer; gs_* calls never allocate."""
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
       t0, t1, iterations, lm, ep, motion_only, tables=None):
    """droid.cpp:88-117.  Mutates `poses`/`disps`; returns [dx, dz] (dz None if motion_only).

    `tables` (optional, not in the reference): a dict the caller keeps per EDGE SET.  It then owns the call's workspace,
    and a second call with the same ii / jj tensors, window and depth-row count skips the index-table kernel
    (GS_BA_REUSE_TABLES) -- FactorGraph.update runs six calls per keyframe on one edge set."""
    _chk("targets", targets, torch.float32)
    _chk("weights", weights, torch.float32)
    _chk("poses", poses, torch.float32)
    _chk("disps", disps, torch.float32)
    _chk("intrinsics", intrinsics, torch.float32)
    _chk("disps_sens", disps_sens, torch.float32)
    _chk("ii", ii, torch.int64)
    _chk("jj", jj, torch.int64)
    if not eta.is_contiguous():       # the reference does not check eta; it views it (-1, ht*wd)
        eta = eta.contiguous()
    dev = poses.device
    nbuf, ht, wd = disps.shape
    E = ii.shape[0]
    P = int(t1) - int(t0)
    hw = ht * wd
    M = eta.reshape(-1, hw).shape[0]
    L = _lib.lib()
    need = L.gs_ba_workspace_bytes(E, P, M, nbuf, hw)
    flags = 0
    new_key = None
    if tables is None:
        ws = _workspace(dev, need + 256)
    else:
        key = (ii.data_ptr(), ii._version, jj.data_ptr(), jj._version, E, int(t0), int(t1), M, nbuf, ht, wd, dev)
        ws = tables.get("workspace")
        if ws is not None and tables.get("key") == key and ws.numel() >= need + 256:
            flags = 1                                        # GS_BA_REUSE_TABLES
        else:
            ws = tables["workspace"] = torch.empty(need + 256, dtype=torch.uint8, device=dev)
            tables.pop("key", None)                          # set only once the tables have actually been built
            new_key = key
    dx = torch.empty(P, 6, dtype=torch.float32, device=dev)
    # (depth rows the kernels skip -- only when `eta` has more rows than the graph has depth keyframes, status [1] -- are
    # zeroed by ba_update_kernel itself; with zero iterations nothing runs, hence the explicit zeros then)
    dz = None if motion_only else (torch.empty if int(iterations) > 0 else torch.zeros)(M, hw, dtype=torch.float32,
                                                                                      device=dev)
    status = _ba_status.get(dev.index)
    if status is None:
        status = _ba_status[dev.index] = torch.zeros(4, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = L.gs_ba_ex(_lib.ptr(poses), _lib.ptr(disps), _lib.ptr(intrinsics), _lib.ptr(disps_sens),
                        _lib.ptr(targets), _lib.ptr(weights), _lib.ptr(eta), _lib.ptr(ii), _lib.ptr(jj),
                        int(t0), int(t1), int(iterations), float(lm), float(ep), int(bool(motion_only)),
                        E, M, nbuf, ht, wd, _lib.ptr(dx), _lib.ptr(dz), _lib.ptr(status),
                        _lib.ptr(ws), ws.numel(), flags, _lib.stream_ptr(dev))
    _lib.check(rc, "droid_backends.ba")
    if new_key is not None:                                  # only a call that went through owns valid tables
        tables["key"] = new_key
        tables["ii"], tables["jj"] = ii, jj                  # keep the addresses in the key alive
    if BA_CHECK:
        st = ba_status(dev)
        if st["depth_rows_mismatch"]:
            raise RuntimeError(f"droid_backends.ba: eta has {M} rows but the graph has {st['depth_keyframes']} depth "
                               "keyframes (unique(arange(t0, t1) U ii))")
    return [dx, dz]


def ba_status(device=None):
    """Status of the last ba() enqueued on `device` (synchronises): number of depth keyframes the kernels found, whether
    that differed from eta's row count (rows beyond it were skipped / dz left zero), Cholesky failures (dx = 0 for those
    iterations, as the reference does).  None if ba() has not run there."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    status = _ba_status.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if status is None:
        return None
    s = status.tolist()
    return {"depth_keyframes": s[0], "depth_rows_mismatch": bool(s[1]), "cholesky_failures": s[2]}


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """droid.cpp:120-131."""
    _chk("poses", poses, torch.float32)
    _chk("disps", disps, torch.float32)
    _chk("intrinsics", intrinsics, torch.float32)
    _chk("ii", ii, torch.int64)
    _chk("jj", jj, torch.int64)
    n = ii.shape[0]
    _, ht, wd = disps.shape
    dist = torch.empty(n, dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        rc = _lib.lib().gs_frame_distance(_lib.ptr(poses), _lib.ptr(disps), _lib.ptr(intrinsics), _lib.ptr(ii),
                                          _lib.ptr(jj), _lib.ptr(dist), n, ht, wd, float(beta),
                                          _lib.stream_ptr(poses.device))
    _lib.check(rc, "droid_backends.frame_distance")
    return dist


def projmap(poses, disps, intrinsics, ii, jj):
    """droid.cpp:133-140 -> [coords [n,h,w,3], valid [n,h,w,1]]."""
    _chk("poses", poses, torch.float32)
    _chk("disps", disps, torch.float32)
    _chk("intrinsics", intrinsics, torch.float32)
    _chk("ii", ii, torch.int64)
    _chk("jj", jj, torch.int64)
    n = ii.shape[0]
    _, ht, wd = disps.shape
    coords = torch.empty(n, ht, wd, 3, dtype=torch.float32, device=poses.device)
    valid = torch.empty(n, ht, wd, 1, dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        rc = _lib.lib().gs_projmap(_lib.ptr(poses), _lib.ptr(disps), _lib.ptr(intrinsics), _lib.ptr(ii),
                                   _lib.ptr(jj), _lib.ptr(coords), _lib.ptr(valid), n, ht, wd,
                                   _lib.stream_ptr(poses.device))
    _lib.check(rc, "droid_backends.projmap")
    return [coords, valid]


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """droid.cpp (depth_filter) -> count [n,h,w]."""
    _chk("poses", poses, torch.float32)
    _chk("disps", disps, torch.float32)
    _chk("intrinsics", intrinsics, torch.float32)
    _chk("ix", ix, torch.int64)
    _chk("thresh", thresh, torch.float32)
    num, ht, wd = disps.shape
    n = ix.shape[0]
    counter = torch.empty(n, ht, wd, dtype=torch.float32, device=disps.device)
    with torch.cuda.device(disps.device):
        rc = _lib.lib().gs_depth_filter(_lib.ptr(poses), _lib.ptr(disps), _lib.ptr(intrinsics), _lib.ptr(ix),
                                        _lib.ptr(thresh), _lib.ptr(counter), n, num, ht, wd,
                                        _lib.stream_ptr(disps.device))
    _lib.check(rc, "droid_backends.depth_filter")
    return counter


def iproj(poses, disps, intrinsics):
    """droid.cpp:141-147 -> points [n,h,w,3]."""
    _chk("poses", poses, torch.float32)
    _chk("disps", disps, torch.float32)
    _chk("intrinsics", intrinsics, torch.float32)
    n, ht, wd = disps.shape
    points = torch.empty(n, ht, wd, 3, dtype=torch.float32, device=disps.device)
    with torch.cuda.device(disps.device):
        rc = _lib.lib().gs_iproj(_lib.ptr(poses), _lib.ptr(disps), _lib.ptr(intrinsics), _lib.ptr(points),
                                 n, ht, wd, _lib.stream_ptr(disps.device))
    _lib.check(rc, "droid_backends.iproj")
    return points


def corr_index_forward(volume, coords, radius):
    """droid.cpp:149-158 -> [corr [n,2r+1,2r+1,h1,w1]] in volume's dtype."""
    _chk("volume", volume)
    _chk("coords", coords, torch.float32)
    if volume.dtype not in _DT:
        raise RuntimeError(f"corr_index_forward: unsupported dtype {volume.dtype}")
    n, h1, w1, h2, w2 = volume.shape
    rd = 2 * radius + 1
    corr = torch.empty(n, rd, rd, h1, w1, dtype=volume.dtype, device=volume.device)
    with torch.cuda.device(volume.device):
        rc = _lib.lib().gs_corr_index_forward(_lib.ptr(volume), _lib.ptr(coords), _lib.ptr(corr), n, h1, w1, h2,
                                              w2, int(radius), _DT[volume.dtype],
                                              _lib.stream_ptr(volume.device))
    _lib.check(rc, "droid_backends.corr_index_forward")
    return [corr]


def corr_index_backward(volume, coords, corr_grad, radius):
    """droid.cpp:160-171 -> [volume_grad]."""
    _chk("volume", volume)
    _chk("coords", coords, torch.float32)
    _chk("corr_grad", corr_grad, volume.dtype)
    n, h1, w1, h2, w2 = volume.shape
    grad = torch.zeros_like(volume)
    with torch.cuda.device(volume.device):
        rc = _lib.lib().gs_corr_index_backward(_lib.ptr(coords), _lib.ptr(corr_grad), _lib.ptr(grad), n, h1, w1,
                                               h2, w2, int(radius), _DT[volume.dtype],
                                               _lib.stream_ptr(volume.device))
    _lib.check(rc, "droid_backends.corr_index_backward")
    return [grad]


CORR_ROWMAJOR, CORR_TILE8 = 0, 1


def corr_lookup_pyramid(pyramid, coords, radius=3, channels_last=False, layout=CORR_ROWMAJOR, map_size=None):
    """Fused 4-level form of CorrBlock.__call__ (src/modules/corr.py:43-53): one launch.
    pyramid: 4 tensors [n,h1,w1,h2>>l,w2>>l]; coords f32 [n,h1,w1,2] -> [n,196,h1,w1]
    (memory format torch.channels_last when `channels_last`).  With layout=CORR_TILE8 (fp16,
    channels_last) levels 0-1 are the [n,h1,w1,plane] tile8 tensors of corr_volume_pyramid and
    map_size=(h2, w2) must be given."""
    assert len(pyramid) == 4
    for i, v in enumerate(pyramid):
        _chk(f"pyramid[{i}]", v, pyramid[0].dtype)
    _chk("coords", coords, torch.float32)
    if layout == CORR_TILE8:
        n, h1, w1 = pyramid[0].shape[:3]
        h2, w2 = map_size
        for l in range(4):
            want = (n, h1, w1, _lib.lib().gs_corr_level_elems(h2, w2, l, layout)) if l < 2 else (n, h1, w1, h2 >> l, w2 >> l)
            if tuple(pyramid[l].shape) != want:
                raise RuntimeError(f"tile8 pyramid level {l} has shape {tuple(pyramid[l].shape)}, expected {want}")
    else:
        n, h1, w1, h2, w2 = pyramid[0].shape
        for l in range(4):
            if tuple(pyramid[l].shape) != (n, h1, w1, h2 >> l, w2 >> l):
                raise RuntimeError(f"pyramid level {l} has shape {tuple(pyramid[l].shape)}")
    rd = 2 * radius + 1
    corr = torch.empty((n, 4 * rd * rd, h1, w1), dtype=pyramid[0].dtype, device=coords.device,
                       memory_format=torch.channels_last if channels_last else torch.contiguous_format)
    with torch.cuda.device(coords.device):
        rc = _lib.lib().gs_corr_lookup_pyramid(_lib.ptr(pyramid[0]), _lib.ptr(pyramid[1]), _lib.ptr(pyramid[2]),
                                               _lib.ptr(pyramid[3]), _lib.ptr(coords), _lib.ptr(corr), n, h1, w1,
                                               h2, w2, int(radius), _DT[pyramid[0].dtype], int(bool(channels_last)),
                                               int(layout), _lib.stream_ptr(coords.device))
    _lib.check(rc, "corr_lookup_pyramid")
    return corr


def reproject(poses, disps, intrinsics, ii, jj):
    """Fused DepthVideo.reproject (src/depth_video.py:207-217): coords [1,n,h,w,2], valid [1,n,h,w,1]."""
    _chk("poses", poses, torch.float32)
    _chk("disps", disps, torch.float32)
    _chk("intrinsics", intrinsics, torch.float32)
    _chk("ii", ii, torch.int64)
    _chk("jj", jj, torch.int64)
    n = ii.shape[0]
    _, ht, wd = disps.shape
    coords = torch.empty(1, n, ht, wd, 2, dtype=torch.float32, device=poses.device)
    valid = torch.empty(1, n, ht, wd, 1, dtype=torch.float32, device=poses.device)
    with torch.cuda.device(poses.device):
        rc = _lib.lib().gs_reproject(_lib.ptr(poses), _lib.ptr(disps), _lib.ptr(intrinsics), _lib.ptr(ii),
                                     _lib.ptr(jj), _lib.ptr(coords), _lib.ptr(valid), n, ht, wd,
                                     _lib.stream_ptr(poses.device))
    _lib.check(rc, "reproject")
    return coords, valid


def altcorr_forward(fmap1, fmap2, coords, radius):
    """droid.cpp:173-184: fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords f32 [B,S,H1,W1,2]
    -> [corr [B,S,(2r+1)^2,H1,W1]] in fmap1's dtype."""
    _chk("fmap1", fmap1)
    _chk("fmap2", fmap2, fmap1.dtype)
    _chk("coords", coords, torch.float32)
    if fmap1.dtype not in (torch.float16, torch.float32):
        raise RuntimeError(f"altcorr_forward: unsupported dtype {fmap1.dtype}")
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    rd = 2 * radius + 1
    corr = torch.empty(B, S, rd * rd, H1, W1, dtype=fmap1.dtype, device=fmap1.device)
    with torch.cuda.device(fmap1.device):
        rc = _lib.lib().gs_altcorr_forward(_lib.ptr(fmap1), _lib.ptr(fmap2), _lib.ptr(coords), _lib.ptr(corr), B, S,
                                           H1, W1, H2, W2, C, int(radius), _DT[fmap1.dtype],
                                           _lib.stream_ptr(fmap1.device))
    _lib.check(rc, "droid_backends.altcorr_forward")
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, radius):
    """droid.cpp:186-199 (training path): fp32 tensors -> [fmap1_grad, fmap2_grad, coords_grad]; as in the
    reference, coords_grad is all zeros (altcorr_kernel.cu never writes it)."""
    for name, t in (("fmap1", fmap1), ("fmap2", fmap2), ("coords", coords), ("corr_grad", corr_grad)):
        _chk(name, t, torch.float32)
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    g1 = torch.empty_like(fmap1)
    g2 = torch.zeros_like(fmap2)
    gc = torch.zeros_like(coords)
    with torch.cuda.device(fmap1.device):
        rc = _lib.lib().gs_altcorr_backward(_lib.ptr(fmap1), _lib.ptr(fmap2), _lib.ptr(coords), _lib.ptr(corr_grad),
                                            _lib.ptr(g1), _lib.ptr(g2), B, S, H1, W1, H2, W2, C, int(radius),
                                            _lib.stream_ptr(fmap1.device))
    _lib.check(rc, "droid_backends.altcorr_backward")
    return [g1, g2, gc]


CORR_VOLUME_MAX_W = 96      # corr_volume_kernel: three 32-column tiles per wave (csrc/corr_build.hip, MAXT)


def corr_volume_supported(fmap1):
    """shapes gs_corr_volume_pyramid covers: fp16, 128 channels, map width a multiple of 4 up to 96 -- every reference
    config, EuRoC's 40 x 60 maps (w % 8 == 4: a block of four target rows is 7.5 MFMA column tiles; the kernel runs the
    eighth half-empty, csrc/corr_build.hip) included"""
    n, dim, h, w = fmap1.shape
    return fmap1.dtype == torch.float16 and dim == 128 and h >= 8 and w % 4 == 0 and 8 <= w <= CORR_VOLUME_MAX_W


def corr_tile8_supported(fmap):
    return corr_volume_supported(fmap) and fmap.shape[-1] % 16 == 0


def corr_untile8(vol, hl, wl):
    """[n,h,w,plane] tile8 level -> the reference's [n,h,w,hl,wl] planes (tests / debugging)."""
    n, h, w, plane = vol.shape
    nty, ntx = (hl + 7) // 8, (wl + 7) // 8
    t = vol.view(n, h, w, nty, ntx, 8, 8).permute(0, 1, 2, 3, 5, 4, 6).reshape(n, h, w, nty * 8, ntx * 8)
    return t[..., :hl, :wl].contiguous()


def corr_volume_pyramid(fmap1, fmap2, layout=CORR_ROWMAJOR):
    """CorrBlock.__init__ + CorrBlock.corr (src/modules/corr.py:26-41,67-76) in one pass:
    fmap1, fmap2 f16 [n,128,h,w] -> list of 4 tensors [n,h,w,h>>l,w>>l] (MFMA GEMM fused with the
    three average pools; the volume is written once and never re-read).  layout=CORR_TILE8 returns
    levels 0-1 as [n,h,w,plane] in the lookup-friendly tile layout (include/goslam_hip.h)."""
    _chk("fmap1", fmap1, torch.float16)
    _chk("fmap2", fmap2, torch.float16)
    if fmap1.shape != fmap2.shape or not corr_volume_supported(fmap1):
        raise RuntimeError(f"corr_volume_pyramid: unsupported shape {tuple(fmap1.shape)} / {tuple(fmap2.shape)}")
    n, dim, h, w = fmap1.shape
    dev = fmap1.device
    L = _lib.lib()
    if layout == CORR_TILE8:
        if not corr_tile8_supported(fmap1):
            raise RuntimeError(f"corr_volume_pyramid: tile8 layout needs w % 16 == 0, got {tuple(fmap1.shape)}")
        vols = [torch.empty(n, h, w, L.gs_corr_level_elems(h, w, l, layout), dtype=torch.float16, device=dev)
                for l in range(2)]
        vols += [torch.empty(n, h, w, h >> l, w >> l, dtype=torch.float16, device=dev) for l in (2, 3)]
    else:
        vols = [torch.empty(n, h, w, h >> l, w >> l, dtype=torch.float16, device=dev) for l in range(4)]
    need = L.gs_corr_volume_workspace_bytes(n, dim, h, w)
    ws = _workspace(dev, need + 256)
    with torch.cuda.device(dev):
        rc = L.gs_corr_volume_pyramid(_lib.ptr(fmap1), _lib.ptr(fmap2), *[_lib.ptr(v) for v in vols], n, dim, h, w,
                                      int(layout), _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
    _lib.check(rc, "corr_volume_pyramid")
    return vols
