"""Differentiable CPU restatement of the mapping step (TEST INFRASTRUCTURE, see oracle/__init__.py).

Same mathematics as oracle/neus_oracle.py::neus_forward, written with differentiable torch ops so
that torch.autograd yields the reference gradients of the mapper's losses
(reference src/mapping.py:96-137) with respect to every trained parameter:
grid table, sdf_layer weight/bias, colour `_B`, colour MLP, variance.  The SDF gradient is
written as the explicit formula (W0[:3] + 1/2 sum g_lf dy_dx_lf) * inside * 2/span, whose
dependence on the grid entries and on W0 is differentiable -- this is what the reference obtains
through `autograd.grad(..., create_graph=True)` (InstantNeuS.py:141-148) plus tiny-cuda-nn's
double-backward kernels.  fp16 roundings are applied with straight-through estimators.
"""
import math

import torch

from . import neus_oracle as NO


def _ste_half(x):
    return x + (x.to(torch.float16).to(x.dtype) - x).detach()


def _index(meta, l, c):
    """tcnn grid_index on int64 tensors c[...,3]."""
    res = int(meta["resolution"][l])
    size = int(meta["size"][l])
    if int(meta["hashed"][l]):
        idx = (c[..., 0] * 1) ^ ((c[..., 1] * 2654435761) & 0xFFFFFFFF) ^ ((c[..., 2] * 805459861) & 0xFFFFFFFF)
        idx = idx & 0xFFFFFFFF
    else:
        idx = (c[..., 0] + c[..., 1] * res + c[..., 2] * res * res) & 0xFFFFFFFF
    return idx % size


def grid_encode_diff(x, grid, meta, x_differentiable=False):
    """x [n,3] in [0,1] (treated as constant unless x_differentiable), grid f32 [total*2] (differentiable,
    used as its fp16 rounding).  Returns enc [n,32] (fp16-rounded, STE) and dydx [n,32,3] (differentiable
    in grid; with x_differentiable also in x: the mixed second derivatives of the trilinear interpolation,
    tiny-cuda-nn's kernel_grid_backward_input_backward_input)."""
    n = x.shape[0]
    g16 = _ste_half(grid).view(-1, 2)
    encs, dys = [], []
    xd = x.detach()
    for l in range(NO.N_LEVELS):
        scale = float(meta["scale"][l])
        off = int(meta["offset"][l])
        pos = (xd.double() * scale + 0.5).float()
        gfl = torch.floor(pos)
        f = pos - gfl
        if x_differentiable:      # same values; d f / d x = scale inside a cell (tcnn's input gradients)
            f = f + (x - xd) * scale
        gi = gfl.to(torch.int64)
        vals = []
        for corner in range(8):
            c = torch.stack([gi[:, d] + ((corner >> d) & 1) for d in range(3)], -1)
            vals.append(g16[_index(meta, l, c) + off])            # [n,2]
        val = 0
        for corner in range(8):
            w = torch.ones(n)
            for d in range(3):
                w = w * (f[:, d] if (corner >> d) & 1 else (1 - f[:, d]))
            val = val + w[:, None] * vals[corner]
        encs.append(val)
        dl = []
        for gd in range(3):
            others = [d for d in range(3) if d != gd]
            acc = 0
            for k in range(4):
                w = torch.full((n,), scale)
                cl = 0
                for b, d in enumerate(others):
                    bit = (k >> b) & 1
                    w = w * (f[:, d] if bit else (1 - f[:, d]))
                    cl |= bit << d
                cr = cl | (1 << gd)
                acc = acc + w[:, None] * (vals[cr] - vals[cl])
            dl.append(acc)                                        # [n,2]
        dys.append(torch.stack(dl, -1))                           # [n,2,3]
    enc = _ste_half(torch.cat(encs, 1))
    dydx = torch.cat(dys, 1)
    return enc, dydx


def mlp_diff(x, params, n_in=67, n_out=3, width=64):
    pad_in = (n_in + 15) // 16 * 16
    pad_out = (n_out + 15) // 16 * 16
    w = _ste_half(params)
    o = 0
    W1 = w[o:o + width * pad_in].view(width, pad_in); o += width * pad_in
    W2 = w[o:o + width * width].view(width, width); o += width * width
    W3 = w[o:o + pad_out * width].view(pad_out, width)
    xin = torch.cat([_ste_half(x), torch.ones(x.shape[0], pad_in - n_in)], 1)
    h = _ste_half(torch.relu(xin @ W1.t()))
    h = _ste_half(torch.relu(h @ W2.t()))
    return _ste_half(h @ W3.t())[:, :n_out]


def neus_forward_diff(rays_o, rays_d, z_vals, dists, P, meta=None):
    """P: dict of (possibly requires_grad) tensors grid, sdf_w, sdf_b, color_B, mlp, variance
    (0-dim tensor) + bound, rt_bound.  Returns the reference's output dict, differentiable."""
    meta = meta or NO.grid_meta()
    n, s = z_vals.shape
    z_mid = z_vals + dists / 2.0
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_mid[:, :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(n, s, 3).reshape(-1, 3)
    mask = NO.in_bound(pts, P["rt_bound"])
    if mask.float().sum() < 1:
        mask[:100] = True
    pm = pts[mask]
    bound = P["bound"]
    span = bound[:, 1] - bound[:, 0]
    p = (pm - bound[:, 0]) / span * 2.0 - 1.0
    inside = ((p >= -1.0) & (p <= 1.0)).float()
    p = p.clamp(-1.0, 1.0)
    enc, dydx = grid_encode_diff((p + 1) / 2, P["grid"], meta)
    out = torch.cat([p, enc], -1) @ P["sdf_w"].t() + P["sdf_b"]
    g_enc = _ste_half(P["sdf_w"][0, 3:])
    g_view = torch.einsum("ncd,c->nd", dydx, g_enc)
    grad_m = (P["sdf_w"][0, :3][None] + g_view / 2) * inside * 2.0 / span
    npts = pts.shape[0]
    sdf = torch.ones(npts, 1) * 100
    grads = torch.zeros(npts, 3)
    feat = torch.zeros(npts, 31)
    sdf = sdf.index_put((mask,), out[:, :1])
    grads = grads.index_put((mask,), grad_m)
    feat = feat.index_put((mask,), out[:, 1:])
    inv_s = torch.exp(P["variance"] * 10.0).clip(1e-6, 1e6)
    alpha = NO.get_alpha(sdf, grads, dirs, dists, inv_s)
    emb = torch.sin(pm @ P["color_B"])
    mlp_in = torch.cat([emb, grad_m, out[:, 1:]], 1)
    o_rgb = _ste_half(torch.sigmoid(mlp_diff(mlp_in, P["mlp"])))
    rgb = torch.zeros(npts, 3).index_put((mask,), o_rgb)
    sdf = sdf.reshape(n, s)
    rgb = rgb.reshape(n, s, 3)
    alpha = (alpha * mask[:, None]).reshape(n, s)
    grads = grads.reshape(n, s, 3)
    m2 = mask.reshape(n, s)
    weights = alpha * torch.cumprod(torch.cat([torch.ones(n, 1), 1 - alpha + 1e-7], 1), 1)[:, :-1]
    depth = (z_mid * weights).sum(1, keepdim=True)
    gerr = (torch.linalg.norm(grads, ord=2, dim=2) - 1.0) ** 2 * m2
    return {
        "color": (rgb * weights[:, :, None]).sum(1), "depth": depth,
        "depth_variance": ((z_mid - depth) ** 2 * weights).sum(1, keepdim=True),
        "normal": ((grads * weights[:, :, None]) * m2[:, :, None]).sum(1),
        "weight_sum": weights.sum(1, keepdim=True), "sdf": sdf, "z_vals": z_mid,
        "gradient_error": gerr.mean().unsqueeze(0),
    }


def mapping_loss(ret, rays_color, rays_depth, truncation=0.16, sparse_factor=5, w_color=2.0, w_sdf=2.0,
                 w_eikonal=0.1, uncertainty=True):
    """Mapper.optimize_map's loss (reference src/mapping.py:96-132)."""
    rd = rays_depth.reshape(-1, 1)
    vm = (rd > 0).reshape(-1)
    rd, rc = rd[vm], rays_color[vm]
    est_c, est_d = ret["color"][vm], ret["depth"][vm]
    sdf, z = ret["sdf"][vm], ret["z_vals"][vm]
    dv = ret["depth_variance"][vm]
    uw = 1.0 / torch.sqrt(dv.detach() + 1e-10) if uncertainty else torch.ones_like(dv)
    total = torch.abs(est_c - rc).mean() * w_color
    total = total + (torch.abs(est_d - rd) * uw).mean()
    e, f = NO.compute_sdf_error(sdf, z, rd, truncation, sparse_factor)
    total = total + (e + f) * w_sdf
    total = total + w_eikonal * ret["gradient_error"].mean()
    return total
