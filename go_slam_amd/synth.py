"""Synthetic RGB-D keyframe graphs / ray batches with the shapes of the BASELINE configs.

There is no network or dataset on the GPU box, so benchmarks and parity tests run on
seeded synthetic inputs shaped like the reference's workloads:
intrinsics of configs/go_slam.yaml:80-83 scaled to the 1/8-resolution maps, keyframe poses on a
smooth arc, smooth depth in [1,4] m, a frontend-like window graph (|i-j|<=2 bidirectional +
random proximity pairs).  Everything is generated on the CPU with a torch.Generator so the
same tensors can be fed to the HIP path and to the CPU oracle.
"""
import math

import torch

SHAPES = {
    "S480": (60, 80, 640.0),     # 480x640 input -> 60x80 maps (BASELINE metric shape)
    "Rep": (40, 80, 640.0),      # Replica 320x640
    "Scan": (30, 40, 320.0),     # ScanNet 240x320
    "tiny": (12, 16, 128.0),     # unit-test shape
}


def _quat_mul(a, b):
    ax, ay, az, aw = a.unbind(-1)
    bx, by, bz, bw = b.unbind(-1)
    return torch.stack([
        aw * bx + ax * bw + ay * bz - az * by,
        aw * by + ay * bw + az * bx - ax * bz,
        aw * bz + az * bw + ax * by - ay * bx,
        aw * bw - ax * bx - ay * by - az * bz,
    ], dim=-1)


def make_intrinsics(shape="S480"):
    ht, wd, w_out = SHAPES[shape]
    s = (w_out / 640.0) / 8.0
    return torch.tensor([577.590698 * s, 578.729797 * s, 318.905426 * s, 242.683609 * s],
                        dtype=torch.float32)


def make_video(num_kf, shape="S480", seed=43, rgbd=True, buffer=None):
    """Returns dict(poses [B,7], disps [B,h,w], disps_sens [B,h,w], intrinsics [B,4])."""
    g = torch.Generator().manual_seed(seed)
    ht, wd, _ = SHAPES[shape]
    B = buffer or num_kf
    poses = torch.zeros(B, 7)
    poses[:, 6] = 1.0
    t = torch.zeros(3)
    q = torch.tensor([0.0, 0.0, 0.0, 1.0])
    for k in range(num_kf):
        if k > 0:
            t = t + torch.randn(3, generator=g) * 0.05
            ang = torch.randn(3, generator=g) * math.radians(2.0)
            th = ang.norm().clamp(min=1e-8)
            dq = torch.cat([torch.sin(th / 2) * ang / th, torch.cos(th / 2)[None]])
            q = _quat_mul(dq, q)
            q = q / q.norm()
        poses[k, :3] = t
        poses[k, 3:] = q
    depth = 1.0 + 3.0 * torch.rand(B, 1, ht, wd, generator=g)
    depth = torch.nn.functional.avg_pool2d(torch.nn.functional.pad(depth, (1, 1, 1, 1), mode="replicate"), 3, 1)
    disps = (1.0 / depth[:, 0]).contiguous()
    if rgbd:
        disps_sens = disps + 0.01 * torch.randn(B, ht, wd, generator=g)
        disps_sens = disps_sens.clamp(min=0.05)
        # a few invalid sensor pixels, as real depth maps have
        disps_sens = torch.where(torch.rand(B, ht, wd, generator=g) < 0.05,
                                 torch.zeros_like(disps_sens), disps_sens)
    else:
        disps_sens = torch.zeros(B, ht, wd)
    intr = make_intrinsics(shape)[None].repeat(B, 1)
    return dict(poses=poses, disps=disps, disps_sens=disps_sens.contiguous(), intrinsics=intr.contiguous())


def make_graph(num_kf, num_edges, seed=43, radius=2):
    """Frontend-like edge set over keyframes [0,num_kf): all |i-j|<=radius pairs (both
    directions) then random extra pairs up to num_edges.  Returns (ii, jj) int64."""
    g = torch.Generator().manual_seed(seed + 1)
    pairs = []
    for i in range(num_kf):
        for j in range(num_kf):
            if i != j and abs(i - j) <= radius:
                pairs.append((i, j))
    have = set(pairs)
    pairs = pairs[:num_edges]
    have = set(pairs)
    guard = 0
    while len(pairs) < num_edges and guard < 100000:
        guard += 1
        i = int(torch.randint(0, num_kf, (1,), generator=g))
        j = int(torch.randint(0, num_kf, (1,), generator=g))
        if i != j and (i, j) not in have:
            have.add((i, j))
            pairs.append((i, j))
    ii = torch.tensor([p[0] for p in pairs], dtype=torch.int64)
    jj = torch.tensor([p[1] for p in pairs], dtype=torch.int64)
    return ii, jj


def make_ba_problem(num_kf=8, num_edges=20, shape="tiny", seed=43, rgbd=True, noise_px=0.5, coords=None):
    """Full input set of `droid_backends.ba` (targets = reprojection + noise).

    `coords` (optional [E,h,w,2]) are the reprojected coordinates to perturb; when None the
    targets are the pixel grid plus a smooth flow, which keeps this module independent of
    any projection code."""
    g = torch.Generator().manual_seed(seed + 2)
    vid = make_video(num_kf, shape, seed, rgbd)
    ii, jj = make_graph(num_kf, num_edges, seed)
    ht, wd, _ = SHAPES[shape]
    E = len(ii)
    if coords is None:
        v, u = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32),
                              indexing="ij")
        coords = torch.stack([u, v], dim=-1)[None].repeat(E, 1, 1, 1)
    target = coords + noise_px * torch.randn(E, ht, wd, 2, generator=g)
    target = target.permute(0, 3, 1, 2).contiguous()
    weight = torch.rand(E, 2, ht, wd, generator=g)
    t0, t1 = 1, num_kf
    kx = torch.unique(torch.cat([torch.arange(t0, t1), ii]))
    eta = 1e-2 * torch.rand(len(kx), ht, wd, generator=g) + 1e-4
    vid.update(dict(ii=ii, jj=jj, target=target, weight=weight, eta=eta, t0=t0, t1=t1))
    return vid


def make_features(num, shape="S480", seed=43, dim=128, dtype=torch.float16):
    g = torch.Generator().manual_seed(seed + 3)
    ht, wd, _ = SHAPES[shape]
    return torch.randn(num, dim, ht, wd, generator=g).to(dtype)


def make_altcorr_chunk(seed=211):
    """Inputs of the one-launch alt-corr fixture (tests/golden/gen_golden.py::gen_reference_altcorr_pyramid and its GPU
    test): a 9-edge chunk over 5 ScanNet-shaped keyframes (30 x 40 maps, fp16 features); edges 0-4 carry a smooth reprojection-like
    flow (the matrix-core path), 5-7 three pixels of per-pixel noise (both paths), edge 8 coordinates far outside the map."""
    ht, wd, _ = SHAPES["Scan"]
    g = torch.Generator().manual_seed(seed)
    fm = (torch.randn(1, 5, 128, ht, wd, generator=g) * 1.5).half()
    ii = torch.tensor([0, 1, 2, 4, 3, 1, 0, 2, 4])
    jj = torch.tensor([1, 0, 4, 2, 3, 3, 4, 0, 1])
    ys, xs = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32), indexing="ij")
    coords = []
    for e in range(9):
        a = 1.0 + 0.15 * (torch.rand(4, generator=g) - 0.5)
        t = 4.0 * torch.randn(2, generator=g)
        x = a[0] * xs + 0.05 * (a[1] - 1) * ys + t[0] + 0.7 * torch.sin(ys / 5.0 + t[1])
        y = a[2] * ys + 0.05 * (a[3] - 1) * xs + t[1] + 0.7 * torch.cos(xs / 6.0 + t[0])
        c = torch.stack([x, y], -1)
        if e >= 5:
            c = c + 3.0 * torch.randn(ht, wd, 2, generator=g)
        if e == 8:
            c = c * 3.0 - 20.0
        coords.append(c)
    return fm, ii, jj, torch.stack(coords)[None].contiguous()
