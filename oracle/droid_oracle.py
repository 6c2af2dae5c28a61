"""CPU restatement of the reference's `droid_backends` module and its Python callers
(TEST INFRASTRUCTURE -- see oracle/__init__.py; never imported by the product path).

File names below are relative to /root/reference/src:
  lib/droid.cpp:237-250          the 9 exported functions (names/signatures kept)
  lib/droid_kernels.cu           BA / geometry kernels + host Schur/solve code
  lib/correlation_kernels.cu     corr_index_forward/backward
  lib/altcorr_kernel.cu          altcorr_forward
  modules/corr.py                CorrBlock / AltCorrBlock host logic
  geom/projective_ops.py         projective_transform (the DepthVideo.reproject path)

Parity status: the CUDA kernels cannot run in the build container and ship no golden
vectors => "parity unpinned" for them; validated by KATs/finite differences (tests/).
All tensors are CPU torch tensors.  Per-pixel arithmetic is fp32 in the reference's
operation order; pixel reductions are carried in fp64 (the reference reduces in fp32 and
assembles/solves in fp64), which is what the stated tolerances are written against.
"""
import math

import torch
import torch.nn.functional as F

from . import se3

MIN_DEPTH = 0.25          # lib/droid_kernels.cu:26
PY_MIN_DEPTH = 0.2        # geom/projective_ops.py:4


# ----------------------------------------------------------------------------------------
# correlation volume + lookup
# ----------------------------------------------------------------------------------------

def corr_volume(fmap1, fmap2):
    """modules/corr.py:67-76 (CorrBlock.corr) as executed under autocast: fp16 operands / 4,
    fp32-accumulated product rounded to fp16.  fmap: [batch, num, dim, ht, wd]."""
    batch, num, dim, ht, wd = fmap1.shape
    dt = fmap1.dtype
    f1 = (fmap1.reshape(batch * num, dim, ht * wd) / 4.0).float()
    f2 = (fmap2.reshape(batch * num, dim, ht * wd) / 4.0).float()
    if dt == torch.float16:   # the division result is rounded to fp16 before the GEMM
        f1 = f1.half().float()
        f2 = f2.half().float()
    corr = torch.matmul(f1.transpose(1, 2), f2).to(dt)
    return corr.view(batch, num, ht, wd, ht, wd)


def corr_pyramid(fmap1, fmap2, num_levels=4):
    """modules/corr.py:26-41 (CorrBlock.__init__): 4-level avg-pool pyramid of the volume.
    Returns a list of [batch*num, h1, w1, h2//2^i, w2//2^i] tensors."""
    corr = corr_volume(fmap1, fmap2)
    batch, num, h1, w1, h2, w2 = corr.shape
    dt = corr.dtype
    corr = corr.reshape(batch * num * h1 * w1, 1, h2, w2)
    pyr = []
    for i in range(num_levels):
        pyr.append(corr.view(batch * num, h1, w1, h2 // 2 ** i, w2 // 2 ** i))
        if i + 1 < num_levels:   # (the reference pools once more and drops the result)
            corr = F.avg_pool2d(corr.float(), kernel_size=2, stride=2).to(dt)
    return pyr


def corr_index_forward(volume, coords, r):
    """lib/correlation_kernels.cu:19-70.  volume [N,h1,w1,h2,w2] (f16/f32/f64), coords
    f32 [N,2,h1,w1].  Returns [corr [N,2r+1,2r+1,h1,w1]] in volume's dtype.

    The arithmetic is carried in the volume's dtype exactly as the kernel does
    (`corr += s * scalar_t(w)`, i outer / j inner), so for fp16 volumes the result is the
    reference's fp16-accumulated value, not an fp32 sum rounded once."""
    N, h1, w1, h2, w2 = volume.shape
    dt = volume.dtype
    rd = 2 * r + 1
    x0 = coords[:, 0]
    y0 = coords[:, 1]
    fx0 = torch.floor(x0)
    fy0 = torch.floor(y0)
    dx = x0 - fx0
    dy = y0 - fy0
    w_nw = (dx * dy).to(dt)                       # scalar_t(dx*dy)
    w_ne = (dx * (1.0 - dy)).to(dt)               # scalar_t(dx*(1-dy))   [i>0, j<rd]
    w_sw = ((1.0 - dx) * dy).to(dt)               # scalar_t((1-dx)*dy)   [i<rd, j>0]
    w_se = ((1.0 - dx) * (1.0 - dy)).to(dt)
    ix0 = fx0.to(torch.int64)
    iy0 = fy0.to(torch.int64)
    flat = volume.reshape(N, h1, w1, h2 * w2)
    corr = torch.zeros(N, rd, rd, h1, w1, dtype=dt)
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = ix0 - r + i
            y1 = iy0 - r + j
            inb = (y1 >= 0) & (y1 < h2) & (x1 >= 0) & (x1 < w2)
            idx = (y1.clamp(0, h2 - 1) * w2 + x1.clamp(0, w2 - 1)).unsqueeze(-1)
            s = torch.gather(flat, 3, idx).squeeze(-1)
            s = torch.where(inb, s, torch.zeros_like(s))
            if i > 0 and j > 0:
                corr[:, i - 1, j - 1] = corr[:, i - 1, j - 1] + s * w_nw
            if i > 0 and j < rd:
                corr[:, i - 1, j] = corr[:, i - 1, j] + s * w_ne
            if i < rd and j > 0:
                corr[:, i, j - 1] = corr[:, i, j - 1] + s * w_sw
            if i < rd and j < rd:
                corr[:, i, j] = corr[:, i, j] + s * w_se
    return [corr]


def corr_index_backward(volume, coords, corr_grad, r):
    """lib/correlation_kernels.cu:73-124: gradient of the lookup wrt the volume."""
    N, h1, w1, h2, w2 = volume.shape
    dt = volume.dtype
    rd = 2 * r + 1
    x0 = coords[:, 0]
    y0 = coords[:, 1]
    fx0 = torch.floor(x0)
    fy0 = torch.floor(y0)
    dx = x0 - fx0
    dy = y0 - fy0
    ix0 = fx0.to(torch.int64)
    iy0 = fy0.to(torch.int64)
    grad = torch.zeros(N, h1, w1, h2 * w2, dtype=dt)
    for i in range(rd + 1):
        for j in range(rd + 1):
            x1 = ix0 - r + i
            y1 = iy0 - r + j
            inb = (y1 >= 0) & (y1 < h2) & (x1 >= 0) & (x1 < w2)
            g = torch.zeros(N, h1, w1, dtype=dt)
            if i > 0 and j > 0:
                g = g + corr_grad[:, i - 1, j - 1] * (dx * dy).to(dt)
            if i > 0 and j < rd:
                g = g + corr_grad[:, i - 1, j] * (dx * (1.0 - dy)).to(dt)
            if i < rd and j > 0:
                g = g + corr_grad[:, i, j - 1] * ((1.0 - dx) * dy).to(dt)
            if i < rd and j < rd:
                g = g + corr_grad[:, i, j] * ((1.0 - dx) * (1.0 - dy)).to(dt)
            g = torch.where(inb, g, torch.zeros_like(g))
            idx = (y1.clamp(0, h2 - 1) * w2 + x1.clamp(0, w2 - 1)).unsqueeze(-1)
            grad.scatter_add_(3, idx, g.unsqueeze(-1))
    return [grad.view(N, h1, w1, h2, w2)]


def corr_lookup(pyramid, coords, radius=3):
    """modules/corr.py:43-53 (CorrBlock.__call__): coords [batch,num,ht,wd,2] ->
    [batch, num, levels*(2r+1)^2, ht, wd]."""
    batch, num, ht, wd, _ = coords.shape
    c = coords.permute(0, 1, 4, 2, 3).contiguous().view(batch * num, 2, ht, wd)
    out = []
    for i, vol in enumerate(pyramid):
        corr, = corr_index_forward(vol, c / 2 ** i, radius)
        out.append(corr.view(batch, num, -1, ht, wd))
    return torch.cat(out, dim=2)


def altcorr_forward(fmap1, fmap2, coords, r):
    """lib/altcorr_kernel.cu:27-149.  fmap1 [B,H1,W1,C], fmap2 [B,H2,W2,C], coords
    f32 [B,S,H1,W1,2] -> [corr [B,S,(2r+1)^2,H1,W1]], channel = iy + rd*ix (x-major)."""
    B, H1, W1, C = fmap1.shape
    _, H2, W2, _ = fmap2.shape
    S = coords.shape[1]
    dt = fmap1.dtype
    rd = 2 * r + 1
    x2 = coords[..., 0]                      # [B,S,H1,W1]
    y2 = coords[..., 1]
    fx = torch.floor(x2)
    fy = torch.floor(y2)
    dx = (x2 - fx)
    dy = (y2 - fy)
    ix0 = fx.to(torch.int64)
    iy0 = fy.to(torch.int64)
    f2flat = fmap2.reshape(B, H2 * W2, C)
    corr = torch.zeros(B, S, rd * rd, H1, W1, dtype=dt)
    f1 = fmap1.unsqueeze(1)                  # [B,1,H1,W1,C]
    for iy in range(rd + 1):
        for ix in range(rd + 1):
            h2 = iy0 - r + iy
            w2 = ix0 - r + ix
            inb = (h2 >= 0) & (h2 < H2) & (w2 >= 0) & (w2 < W2)
            idx = (h2.clamp(0, H2 - 1) * W2 + w2.clamp(0, W2 - 1)).reshape(B, -1)
            g = torch.gather(f2flat, 1, idx.unsqueeze(-1).expand(-1, -1, C)).view(B, S, H1, W1, C)
            s = (f1 * g).sum(-1)
            s = torch.where(inb, s, torch.zeros_like(s))
            if iy > 0 and ix > 0:
                corr[:, :, (iy - 1) + rd * (ix - 1)] += s * (dy * dx).to(dt)
            if iy > 0 and ix < rd:
                corr[:, :, (iy - 1) + rd * ix] += s * (dy * (1 - dx)).to(dt)
            if iy < rd and ix > 0:
                corr[:, :, iy + rd * (ix - 1)] += s * ((1 - dy) * dx).to(dt)
            if iy < rd and ix < rd:
                corr[:, :, iy + rd * ix] += s * ((1 - dy) * (1 - dx)).to(dt)
    return [corr]


def altcorr_backward(fmap1, fmap2, coords, corr_grad, r):
    """lib/altcorr_kernel.cu:151-283,322-354 (fp32, training path): the kernel accumulates exactly the
    vector-Jacobian products of altcorr_forward w.r.t. fmap1 and fmap2 (g = the bilinear-weighted corr_grad
    taps of a window cell; f1_grad += g*f2, f2_grad += g*f1) and never writes coords_grad, which is
    returned as zeros.  Restated as torch.autograd over the forward restatement above."""
    f1 = fmap1.detach().clone().float().requires_grad_(True)
    f2 = fmap2.detach().clone().float().requires_grad_(True)
    (corr,) = altcorr_forward(f1, f2, coords.float(), r)
    g1, g2 = torch.autograd.grad(corr, [f1, f2], corr_grad.float())
    return [g1, g2, torch.zeros_like(coords)]


def altcorr_pyramid(fmaps, num_levels=4):
    """modules/corr.py:98-110 (AltCorrBlock.__init__): fmaps [B,N,C,H,W] -> channels-last
    feature pyramid (features / 4, avg-pooled)."""
    B, N, C, H, W = fmaps.shape
    dt = fmaps.dtype
    f = fmaps.reshape(B * N, C, H, W) / 4.0
    pyr = []
    for i in range(num_levels):
        pyr.append(f.permute(0, 2, 3, 1).contiguous().view(B, N, H // 2 ** i, W // 2 ** i, C))
        if i + 1 < num_levels:
            f = F.avg_pool2d(f.float(), kernel_size=2, stride=2).to(dt)
    return pyr


def altcorr_lookup(pyramid, coords, ii, jj, radius=3):
    """modules/corr.py:112-145 (AltCorrBlock.corr_fn/__call__) for 5-D coords
    [B,N,H,W,2] -> [B,N,levels*(2r+1)^2,H,W] (fp32, as `corr.py:125` casts)."""
    coords = coords.unsqueeze(-2)                      # [B,N,H,W,1,2]
    B, N, H, W, S, _ = coords.shape
    coords = coords.permute(0, 1, 4, 2, 3, 5)
    out = []
    for i, lvl in enumerate(pyramid):
        f1 = pyramid[0][:, ii]
        f2 = lvl[:, jj]
        ci = (coords / 2 ** i).reshape(B * N, S, H, W, 2).contiguous()
        f1 = f1.reshape((B * N,) + f1.shape[2:]).float()
        f2 = f2.reshape((B * N,) + f2.shape[2:]).float()
        corr, = altcorr_forward(f1, f2, ci, radius)
        out.append(corr.view(B, N, S, -1, H, W).permute(0, 1, 3, 4, 5, 2))
    return torch.cat(out, dim=2).squeeze(-1).contiguous()


# ----------------------------------------------------------------------------------------
# geometry
# ----------------------------------------------------------------------------------------

def _grid(ht, wd):
    v, u = torch.meshgrid(torch.arange(ht, dtype=torch.float32),
                          torch.arange(wd, dtype=torch.float32), indexing='ij')
    return u, v


def _rel_pose_kernel(poses, ii, jj, stereo_override):
    """thread-0 prologue of the BA/geometry kernels (lib/droid_kernels.cu:219-248)."""
    ti, qi = poses[ii, :3], poses[ii, 3:]
    tj, qj = poses[jj, :3], poses[jj, 3:]
    tij, qij = se3.rel_se3(ti, qi, tj, qj)
    if stereo_override:
        st = (ii == jj)
        tij = torch.where(st[:, None], torch.tensor([-0.1, 0.0, 0.0]), tij)
        qij = torch.where(st[:, None], torch.tensor([0.0, 0.0, 0.0, 1.0]), qij)
    return tij, qij


def reproject(poses, disps, intrinsics, ii, jj):
    """depth_video.py:207-217 -> geom/projective_ops.py:114-144 (jacobian=False).
    poses [B,7], disps [B,h,w], intrinsics [B,4]; returns coords [1,E,h,w,2], valid
    [1,E,h,w,1].  Group ops follow lietorch's SE3 (q1 q2, t1 + q1*t2)."""
    ht, wd = disps.shape[-2:]
    u, v = _grid(ht, wd)
    fxi, fyi, cxi, cyi = [intrinsics[ii, k][:, None, None] for k in range(4)]
    fxj, fyj, cxj, cyj = [intrinsics[jj, k][:, None, None] for k in range(4)]
    X = (u - cxi) / fxi
    Y = (v - cyi) / fyi
    X0 = torch.stack([X, Y, torch.ones_like(X), disps[ii]], dim=-1)          # projective_ops.py:26-42
    tinv, qinv = se3.se3_inv(poses[ii, :3], poses[ii, 3:])
    tij, qij = se3.se3_mul(poses[jj, :3], poses[jj, 3:], tinv, qinv)         # :123
    st = (ii == jj)
    tij = torch.where(st[:, None], torch.tensor([-0.1, 0.0, 0.0]), tij)      # :124
    qij = torch.where(st[:, None], torch.tensor([0.0, 0.0, 0.0, 1.0]), qij)
    X1 = se3.act_se3(tij[:, None, None], qij[:, None, None], X0)             # :57
    Z = X1[..., 2]
    Zs = torch.where(Z < 0.5 * PY_MIN_DEPTH, torch.ones_like(Z), Z)          # :93
    x = fxj * (X1[..., 0] / Zs) + cxj
    y = fyj * (X1[..., 1] / Zs) + cyj
    valid = ((X1[..., 2] > PY_MIN_DEPTH) & (X0[..., 2] > PY_MIN_DEPTH)).float()
    return torch.stack([x, y], dim=-1)[None], valid[None, ..., None]


def projmap(poses, disps, intrinsics, ii, jj):
    """lib/droid_kernels.cu:427-516, :1463-1488.  coords [n,h,w,3] (channel 2 unused =0)."""
    ht, wd = disps.shape[-2:]
    u, v = _grid(ht, wd)
    fx, fy, cx, cy = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    tij, qij = _rel_pose_kernel(poses, ii, jj, stereo_override=False)
    Xi = torch.stack([((u - cx) / fx).expand(len(ii), -1, -1), ((v - cy) / fy).expand(len(ii), -1, -1),
                      torch.ones(len(ii), ht, wd), disps[ii]], dim=-1)
    Xj = se3.act_se3(tij[:, None, None], qij[:, None, None], Xi)
    ok = Xj[..., 2] > 0.01
    x = torch.where(ok, fx * (Xj[..., 0] / Xj[..., 2]) + cx, u.expand_as(ok))
    y = torch.where(ok, fy * (Xj[..., 1] / Xj[..., 2]) + cy, v.expand_as(ok))
    coords = torch.stack([x, y, torch.zeros_like(x)], dim=-1)
    valid = (Xj[..., 2] > MIN_DEPTH).float()[..., None]
    return [coords, valid]


def frame_distance(poses, disps, intrinsics, ii, jj, beta):
    """lib/droid_kernels.cu:518-657, :1438-1460."""
    n = len(ii)
    ht, wd = disps.shape[-2:]
    u, v = _grid(ht, wd)
    fx, fy, cx, cy = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    tij, qij = _rel_pose_kernel(poses, ii, jj, stereo_override=False)
    out = torch.zeros(n)
    CH = 2048
    for s in range(0, n, CH):
        e = slice(s, min(n, s + CH))
        ne = e.stop - e.start
        Xi = torch.stack([((u - cx) / fx).expand(ne, -1, -1), ((v - cy) / fy).expand(ne, -1, -1),
                          torch.ones(ne, ht, wd), disps[ii[e]]], dim=-1)
        Xj = se3.act_se3(tij[e, None, None], qij[e, None, None], Xi)
        du = fx * (Xj[..., 0] / Xj[..., 2]) + cx - u
        dv = fy * (Xj[..., 1] / Xj[..., 2]) + cy - v
        d = torch.sqrt(du * du + dv * dv)
        ok = Xj[..., 2] > MIN_DEPTH
        accum = (torch.where(ok, beta * d, torch.zeros_like(d))).double().sum((1, 2))
        valid = (ok.double() * beta).sum((1, 2))
        # translation-only flow (:618-636)
        Xt = Xi[..., :3] + Xi[..., 3:4] * tij[e, None, None]
        du = fx * (Xt[..., 0] / Xt[..., 2]) + cx - u
        dv = fy * (Xt[..., 1] / Xt[..., 2]) + cy - v
        d = torch.sqrt(du * du + dv * dv)
        ok = Xt[..., 2] > MIN_DEPTH
        accum = accum + (torch.where(ok, (1 - beta) * d, torch.zeros_like(d))).double().sum((1, 2))
        valid = valid + (ok.double() * (1 - beta)).sum((1, 2))
        total = float(ht * wd) * (beta + (1 - beta))
        res = torch.where(valid / (total + 1e-8) < 0.75, torch.full_like(accum, 1000.0),
                          accum / valid.clamp(min=1e-30))
        out[e] = res.float()
    return out


def iproj(poses, disps, intrinsics):
    """lib/droid_kernels.cu:779-850, :1518-1541.  poses [n,7] (camera->world in callers),
    disps [n,H,W] -> points [n,H,W,3]."""
    n, ht, wd = disps.shape
    u, v = _grid(ht, wd)
    fx, fy, cx, cy = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    Xi = torch.stack([((u - cx) / fx).expand(n, -1, -1), ((v - cy) / fy).expand(n, -1, -1),
                      torch.ones(n, ht, wd), disps], dim=-1)
    Xj = se3.act_se3(poses[:, None, None, :3], poses[:, None, None, 3:], Xi)
    return Xj[..., :3] / Xj[..., 3:4]


def depth_filter(poses, disps, intrinsics, ix, thresh):
    """lib/droid_kernels.cu:661-775, :1491-1515.  Counts, per pixel of frame ix[b], how many
    of the 6 neighbours {-1,-2,-3,+3,+4,+5} see a consistent depth (any of the 4 bilinear
    corners within thresh, tested in double as the kernel's `1.0/dj` promotes)."""
    num, ht, wd = disps.shape
    nb = len(ix)
    u, v = _grid(ht, wd)
    fx, fy, cx, cy = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    counter = torch.zeros(nb, ht, wd)
    dflat = disps.reshape(num, ht * wd)
    for b in range(nb):
        i = int(ix[b])
        t = float(thresh[b])
        for neigh in range(6):
            j = i - neigh - 1 if neigh < 3 else i + neigh
            if j < 0 or j >= num:
                continue
            tij, qij = _rel_pose_kernel(poses, torch.tensor([i]), torch.tensor([j]), False)
            Xi = torch.stack([(u - cx) / fx, (v - cy) / fy, torch.ones(ht, wd), disps[i]], dim=-1)
            Xj = se3.act_se3(tij[0], qij[0], Xi)
            uj = fx * (Xj[..., 0] / Xj[..., 2]) + cx
            vj = fy * (Xj[..., 1] / Xj[..., 2]) + cy
            dj = Xj[..., 3] / Xj[..., 2]
            u0 = torch.floor(uj)
            v0 = torch.floor(vj)
            inb = (u0 >= 0) & (v0 >= 0) & (u0 < wd - 1) & (v0 < ht - 1)
            u0i = u0.clamp(0, wd - 2).to(torch.int64)
            v0i = v0.clamp(0, ht - 2).to(torch.int64)
            base = v0i * wd + u0i
            d00 = dflat[j][base]
            d01 = dflat[j][base + 1]
            d10 = dflat[j][base + wd]
            d11 = dflat[j][base + wd + 1]
            inv = 1.0 / dj.double()
            hit = ((inv - 1.0 / d00.double()).abs() < t) | ((inv - 1.0 / d01.double()).abs() < t) | \
                  ((inv - 1.0 / d10.double()).abs() < t) | ((inv - 1.0 / d11.double()).abs() < t)
            counter[b] += (hit & inb).float()
    return counter


# ----------------------------------------------------------------------------------------
# dense bundle adjustment
# ----------------------------------------------------------------------------------------

def ba_edge_terms(poses, disps, intrinsics, targets, weights, ii, jj):
    """lib/droid_kernels.cu:176-424 (projective_transform_kernel), vectorised over edges.

    Returns per-edge Hs [4,E,6,6] (f64 pixel sums), vs [2,E,6] (f64), and per-pixel fp32
    Eii,Eij [E,6,HW], Cii,bz [E,HW]."""
    E = len(ii)
    ht, wd = disps.shape[-2:]
    HW = ht * wd
    u, v = _grid(ht, wd)
    fx, fy, cx, cy = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    tij, qij = _rel_pose_kernel(poses, ii, jj, stereo_override=True)
    st = (ii == jj)[:, None]

    Xi = torch.stack([((u - cx) / fx).expand(E, -1, -1), ((v - cy) / fy).expand(E, -1, -1),
                      torch.ones(E, ht, wd), disps[ii]], dim=-1).reshape(E, HW, 4)
    Xj = se3.act_se3(tij[:, None], qij[:, None], Xi)
    x, y, z, h = Xj.unbind(-1)
    close = z < MIN_DEPTH
    d = torch.where(close, torch.zeros_like(z), 1.0 / z)
    d2 = d * d
    tg = targets.reshape(E, 2, HW)
    wt = weights.reshape(E, 2, HW)
    wu = torch.where(close, torch.zeros_like(z), 0.001 * wt[:, 0])
    wv = torch.where(close, torch.zeros_like(z), 0.001 * wt[:, 1])
    ru = tg[:, 0] - (fx * d * x + cx)
    rv = tg[:, 1] - (fy * d * y + cy)

    def row(Jj, Jz, w, r):
        C = w * Jz * Jz
        b = w * r * Jz
        w = torch.where(st, torch.zeros_like(w), w)                # :323 / :356
        Ji = -se3.adj_se3(tij[:, None], qij[:, None], Jj)          # :325-326
        Jx = torch.cat([Ji, Jj], dim=-1).double()                  # [E,HW,12]
        wd_ = w.double()
        H = torch.einsum('epn,epm->enm', Jx * wd_[..., None], Jx)
        vv = torch.einsum('epn,ep->en', Jx, (w * r).double())
        Ei = (w * Jz)[..., None] * Ji
        Ej = (w * Jz)[..., None] * Jj
        return C, b, H, vv, Ei, Ej

    Jj_u, Jz_u, Jj_v, Jz_v = _jacobian_rows(fx, fy, x, y, h, d, d2, tij)

    Cu, bu, Hu, vu, Eiu, Eju = row(Jj_u, Jz_u, wu, ru)
    Cv, bv, Hv, vv, Eiv, Ejv = row(Jj_v, Jz_v, wv, rv)
    H = Hu + Hv
    vs_ = vu + vv
    Hs = torch.stack([H[:, :6, :6], H[:, :6, 6:], H[:, 6:, :6], H[:, 6:, 6:]], dim=0)
    vs = torch.stack([vs_[:, :6], vs_[:, 6:]], dim=0)
    Eii = (Eiu + Eiv).permute(0, 2, 1).contiguous()
    Eij = (Eju + Ejv).permute(0, 2, 1).contiguous()
    return Hs, vs, Eii, Eij, Cu + Cv, bu + bv


def _jacobian_rows(fx, fy, x, y, h, d, d2, tij):
    """lib/droid_kernels.cu:312-319,345-352: pose-j rows and depth column of the 2x(6|1) Jacobian."""
    o = torch.zeros_like(x)
    Jj_u = torch.stack([fx * (h * d), fx * o, fx * (-x * h * d2), fx * (-x * y * d2),
                        fx * (1 + x * x * d2), fx * (-y * d)], dim=-1)
    Jz_u = fx * (tij[:, None, 0] * d - tij[:, None, 2] * (x * d2))
    Jj_v = torch.stack([fy * o, fy * (h * d), fy * (-y * h * d2), fy * (-1 - y * y * d2),
                        fy * (x * y * d2), fy * (x * d)], dim=-1)
    Jz_v = fy * (tij[:, None, 1] * d - tij[:, None, 2] * (y * d2))
    return Jj_u, Jz_u, Jj_v, Jz_v


def edge_jacobians(poses, disps, intrinsics, ii, jj):
    """Raw per-pixel Jacobians of the kernel's residual (no weights): returns
    (Ji [E,HW,2,6], Jj [E,HW,2,6], Jz [E,HW,2], z [E,HW]) for cross-checks against the
    reference's Python formulation (geom/projective_ops.py:114-144 with jacobian=True)."""
    E = len(ii)
    ht, wd = disps.shape[-2:]
    HW = ht * wd
    u, v = _grid(ht, wd)
    fx, fy, cx, cy = intrinsics[0], intrinsics[1], intrinsics[2], intrinsics[3]
    tij, qij = _rel_pose_kernel(poses, ii, jj, stereo_override=True)
    Xi = torch.stack([((u - cx) / fx).expand(E, -1, -1), ((v - cy) / fy).expand(E, -1, -1),
                      torch.ones(E, ht, wd), disps[ii]], dim=-1).reshape(E, HW, 4)
    Xj = se3.act_se3(tij[:, None], qij[:, None], Xi)
    x, y, z, h = Xj.unbind(-1)
    d = torch.where(z < MIN_DEPTH, torch.zeros_like(z), 1.0 / z)
    Jj_u, Jz_u, Jj_v, Jz_v = _jacobian_rows(fx, fy, x, y, h, d, d * d, tij)
    Ji_u = -se3.adj_se3(tij[:, None], qij[:, None], Jj_u)
    Ji_v = -se3.adj_se3(tij[:, None], qij[:, None], Jj_v)
    return (torch.stack([Ji_u, Ji_v], 2), torch.stack([Jj_u, Jj_v], 2), torch.stack([Jz_u, Jz_v], 2), z)


def _accum(data, ix, jx):
    """lib/droid_kernels.cu:854-874,948-998: out[j] = sum_{i: ix[i]==jx[j]} data[i]."""
    out = torch.zeros(len(jx), data.shape[1], dtype=torch.float64)
    pos = {int(k): n for n, k in enumerate(jx.tolist())}
    sel = [(n, pos[int(k)]) for n, k in enumerate(ix.tolist()) if int(k) in pos]
    if sel:
        src = torch.tensor([s[0] for s in sel])
        dst = torch.tensor([s[1] for s in sel])
        out.index_add_(0, dst, data[src].double())
    return out.float()


def _solve(H, b, lm, ep):
    """lib/droid_kernels.cu:1192-1213 (SparseBlock::solve): LM damping, fp64 LLT, zeros on
    failure."""
    n = H.shape[0]
    L = H.clone()
    dg = torch.diagonal(L)
    dg += ep + lm * dg.clone()
    Lc, info = torch.linalg.cholesky_ex(L)
    if int(info) != 0:
        return torch.zeros(n, dtype=torch.float32), False
    x = torch.cholesky_solve(b[:, None], Lc)[:, 0]
    return x.float(), True


def ba(poses, disps, intrinsics, disps_sens, targets, weights, eta, ii, jj,
       t0, t1, iterations, lm, ep, motion_only, return_system=False, evt_quirk=True):
    """lib/droid_kernels.cu:1314-1434 (ba_cuda) incl. :1117-1311 (SparseBlock, schur_block).

    Mutates `poses` and `disps` in place, returns [dx [P,6], dz [M,HW] or None].
    Indices outside the window [t0,t1) are dropped from the pose system (reference: <0
    dropped at :1146/:1167; >=P is out-of-range behaviour there and never occurs in callers).
    """
    E = len(ii)
    ht, wd = disps.shape[-2:]
    HW = ht * wd
    P = t1 - t0
    ts = torch.arange(t0, t1)
    ii_exp = torch.cat([ts, ii])
    jj_exp = torch.cat([ts, jj])
    kx, kk_exp = torch.unique(ii_exp, sorted=True, return_inverse=True)
    dx = dz = None
    system = None
    for _ in range(iterations):
        Hs, vs, Eii, Eij, Cii, bz = ba_edge_terms(poses, disps, intrinsics, targets, weights, ii, jj)

        # pose x pose block (:1376-1383)
        H = torch.zeros(P, 6, P, 6, dtype=torch.float64)
        b = torch.zeros(P, 6, dtype=torch.float64)
        # the reference converts each fp32 block to fp64 before summation
        Hs32 = Hs.float().double()
        vs32 = vs.float().double()
        rows = torch.cat([ii, ii, jj, jj]) - t0
        cols = torch.cat([ii, jj, ii, jj]) - t0
        blocks = Hs32.reshape(-1, 6, 6)
        for n in range(4 * E):
            i, j = int(rows[n]), int(cols[n])
            if 0 <= i < P and 0 <= j < P:
                H[i, :, j, :] += blocks[n]
        vrows = torch.cat([ii, jj]) - t0
        vblocks = vs32.reshape(-1, 6)
        for n in range(2 * E):
            i = int(vrows[n])
            if 0 <= i < P:
                b[i] += vblocks[n]

        if motion_only:
            dxv, ok = _solve(H.reshape(6 * P, 6 * P), b.reshape(-1), lm, ep)
            dx = dxv.view(P, 6)
            system = (H.clone(), b.clone())
        else:
            alpha = 0.05                                                     # :1396
            m = (disps_sens[kx] > 0).float().view(-1, HW)
            C = _accum(Cii, ii, kx) + m * alpha + (1 - m) * eta.reshape(-1, HW)
            w = _accum(bz, ii, kx) - m * alpha * (disps[kx] - disps_sens[kx]).view(-1, HW)
            Q = 1.0 / C
            Ei = _accum(Eii.view(E, 6 * HW), ii, ts).view(P, 6, HW)
            Eall = torch.cat([Ei, Eij], dim=0)                               # [P+E,6,HW]

            # Schur complement (:1222-1311)
            pose_n = (jj_exp - t0)
            inwin = (pose_n >= 0) & (pose_n < P)
            for k in range(len(kx)):
                ent = torch.nonzero((kk_exp == k) & inwin).flatten()
                if len(ent) == 0:
                    continue
                Ek = Eall[ent]                                               # [n,6,HW]
                EQ = (Ek * Q[k]).double()                                    # ei = E*q (fp32)
                S = torch.einsum('aip,bjp->aibj', EQ, Ek.double())
                S = S.float().double()                                       # fp32 blocks
                for a in range(len(ent)):
                    for c in range(len(ent)):
                        H[int(pose_n[ent[a]]), :, int(pose_n[ent[c]]), :] -= S[a, :, c, :]
            # rhs: v[n] = sum_pix E_n * (Q*w)[kk[n]]  (:1059-1093, update_rhs :1308)
            qw = Q * w
            vS = torch.einsum('nip,np->ni', Eall.double(), qw[kk_exp].double()).float().double()
            for n in range(len(jj_exp)):
                i = int(pose_n[n])
                if 0 <= i < P:
                    b[i] -= vS[n]
            system = (H.clone(), b.clone())
            dxv, ok = _solve(H.reshape(6 * P, 6 * P), b.reshape(-1), lm, ep)
            dx = dxv.view(P, 6)

            # back-substitution (:1408-1417) with the EvT `<=0` quirk (:1105)
            keep = ((pose_n > 0) if evt_quirk else (pose_n >= 0)) & (pose_n < P)   # evt_quirk=False: tests only
            dxn = dx[pose_n.clamp(0, P - 1)]                                 # [P+E,6]
            dw = torch.einsum('nip,ni->np', Eall, dxn)
            dw = torch.where(keep[:, None], dw, torch.zeros_like(dw))
            dz = Q * (w - _accum(dw, ii_exp, kx))

        # retractions (:877-946)
        tn, qn = se3.retr_se3(dx, poses[t0:t1, :3], poses[t0:t1, 3:])
        poses[t0:t1, :3] = tn
        poses[t0:t1, 3:] = qn
        if not motion_only:
            disps[kx] = disps[kx] + dz.view(-1, ht, wd)
    if return_system:
        return [dx, dz], system
    return [dx, dz]
