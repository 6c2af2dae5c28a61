"""CPU restatement of the mapping hot path (TEST INFRASTRUCTURE -- see oracle/__init__.py).

Follows, relative to /root/reference/src:
  render.py:73-175          Renderer.render_batch_ray (far bound, sample placement, sort, dists)
  InstantNeuS.py:12-32      normalized_3d_coordinate
  InstantNeuS.py:35-94      Encoding (tcnn HashGrid + xyz)
  InstantNeuS.py:97-159     SDFNetwork (Linear 35->32, sdf gradient by autograd)
  InstantNeuS.py:162-205    ColorNetwork (sin(x@B) | normals | feat -> tcnn FullyFusedMLP -> sigmoid)
  InstantNeuS.py:276-370    get_alpha + forward (compositing)
  InstantNeuS.py:372-400    compute_sdf_error

tiny-cuda-nn is an external, un-vendored, unpinned dependency (README.md:95 installs it from git
master; call sites InstantNeuS.py:62,192) and is not installed here, so its two modules are
restated from the published algorithm: multi-resolution hash grid
(16 levels x 2 features, T=2^19, base 16, per-level scale 1.447269237440378, linear
interpolation, fp16 parameters/outputs, fp32 interpolation) and FullyFusedMLP
(67->(80 padded with ones)->64->64->(16 padded)->3, ReLU, no bias, fp16 weights/activations).
**Parity unpinned** for these two; everything in PyTorch in the reference (alpha, compositing,
sampling, losses) is pinned by golden vectors generated from the reference's own code
(tests/golden/).
"""
import math

import numpy as np
import torch

N_LEVELS = 16
N_FEATS = 2
LOG2_T = 19
BASE_RES = 16
PER_LEVEL_SCALE = 1.447269237440378
PRIMES = (1, 2654435761, 805459861)


def grid_meta():
    """tcnn GridEncoding constructor arithmetic in fp32 (grid.h: grid_scale / grid_resolution /
    params_in_level).  Returns dict of numpy arrays: scale f32[16], resolution, size, offset u32."""
    # log2f / exp2f as correctly-rounded fp32 functions (what glibc's libm gives tcnn's host
    # code): evaluate in double, round once to fp32.  numpy's vectorised float32 exp2 is 1 ulp
    # off at level 3, which would shift every sample position of that level.
    log2_s = np.float32(math.log2(float(np.float32(PER_LEVEL_SCALE))))
    scale = np.zeros(N_LEVELS, np.float32)
    res = np.zeros(N_LEVELS, np.uint32)
    size = np.zeros(N_LEVELS, np.uint32)
    off = np.zeros(N_LEVELS, np.uint32)
    hashed = np.zeros(N_LEVELS, np.uint32)
    total = 0
    for l in range(N_LEVELS):
        e = np.float32(2.0 ** float(np.float32(l) * log2_s))
        s = np.float32(e * np.float32(BASE_RES) - np.float32(1.0))
        r = int(np.ceil(s)) + 1
        dense = r ** 3
        n = min(dense, 2 ** 32 - 1)
        n = (n + 7) // 8 * 8
        n = min(n, 1 << LOG2_T)
        scale[l], res[l], size[l], off[l] = s, r, n, total
        hashed[l] = 1 if dense > n else 0          # grid_index: hashmap_size < stride after 3 dims
        total += n
    return dict(scale=scale, resolution=res, size=size, offset=off, hashed=hashed, total=total)


def _grid_index(meta, l, cx, cy, cz):
    """tcnn grid_index<3>: dense stride walk, hash when the level does not fit, then modulo."""
    res = int(meta["resolution"][l])
    size = int(meta["size"][l])
    c = [cx.astype(np.uint64), cy.astype(np.uint64), cz.astype(np.uint64)]
    stride, idx = 1, np.zeros_like(c[0])
    for d in range(3):
        if stride > size:
            break
        idx = (idx + c[d] * stride) & 0xFFFFFFFF
        stride *= res
    if size < stride:
        idx = ((c[0] * PRIMES[0]) ^ (c[1] * PRIMES[1]) ^ (c[2] * PRIMES[2])) & 0xFFFFFFFF
    return (idx % size).astype(np.int64)


# How the 8 corner values of a level are summed / how a layer's dot products are accumulated.  tiny-cuda-nn is absent from
# /root/reference and unpinned, so WHICH of these the installed binary did cannot be checked here:
#   "float"        (default; what the HIP kernels follow) fp32 fused multiply-adds, ONE rounding to fp16 at the end;
#   "half"         upstream's types as published: kernel_grid keeps `result` in vector_t<T = __half> and adds each corner with a
#                  half fma (weight rounded to half, one rounding per corner); FullyFusedMLP keeps wmma accumulator fragments
#                  in half (one rounding per 16-wide k-step, products and the 16-term sum exact inside the matrix unit);
#   "half_mul_add" the older kernel_grid form `result += (T)(weight * (float)val)`: the product rounded to half, then a half add.
# tests/test_oracle_pinned.py::test_half_accumulation_* measures how far the three are apart (profiles/r06_tcnn_half_accumulation.json).
ACCUMULATE = "float"


def _h(a):
    """round an fp32 / fp64 numpy array to fp16 and back (round-to-nearest-even, numpy's conversion)"""
    return a.astype(np.float16).astype(np.float32)


def grid_encode(x, params, meta=None, want_grad=False, accumulate=None):
    """tcnn kernel_grid<T=half,3,2> forward (+ dy_dx): x f32 [n,3] in [0,1]; params f16 (or f32
    master, cast to f16) [total*2].  Returns enc f16 [n,32] and (optionally) dy_dx f32 [n,32,3] (dy_dx is accumulated in
    fp32 upstream too: `vector_fullp_t` gradients).  `accumulate`: see ACCUMULATE above."""
    accumulate = accumulate or ACCUMULATE
    assert accumulate in ("float", "half", "half_mul_add")
    meta = meta or grid_meta()
    n = x.shape[0]
    p16 = params.detach().to(torch.float16).float().numpy().reshape(-1, N_FEATS)
    xn = x.detach().float().numpy()
    enc = np.zeros((n, N_LEVELS * N_FEATS), np.float32)
    dydx = np.zeros((n, N_LEVELS * N_FEATS, 3), np.float32) if want_grad else None
    for l in range(N_LEVELS):
        scale = np.float32(meta["scale"][l])
        off = int(meta["offset"][l])
        # pos = fmaf(scale, x, 0.5): one rounding
        pos = (xn.astype(np.float64) * np.float64(scale) + 0.5).astype(np.float32)
        g = np.floor(pos)
        f = (pos - g).astype(np.float32)
        gi = g.astype(np.int64).astype(np.uint32)
        res = np.zeros((n, N_FEATS), np.float32)
        for corner in range(8):
            w = np.ones(n, np.float32)
            cc = []
            for d in range(3):
                if (corner >> d) & 1:
                    w = (w * f[:, d]).astype(np.float32)
                    cc.append(gi[:, d] + np.uint32(1))
                else:
                    w = (w * (np.float32(1) - f[:, d])).astype(np.float32)
                    cc.append(gi[:, d])
            idx = _grid_index(meta, l, cc[0], cc[1], cc[2]) + off
            val = p16[idx]
            if accumulate == "float":      # result = fmaf(w, val, result)
                res = (w[:, None].astype(np.float64) * val.astype(np.float64) + res.astype(np.float64)).astype(np.float32)
            elif accumulate == "half":     # result = __hfma((half)w, val, result): exact in fp64 (11 + 11 bit product), one rounding
                res = _h(_h(w)[:, None].astype(np.float64) * val.astype(np.float64) + res.astype(np.float64))
            else:                          # result += (half)(w * (float)val)
                res = _h(_h(w[:, None] * val).astype(np.float64) + res.astype(np.float64))
        enc[:, 2 * l:2 * l + 2] = res
        if want_grad:
            for gd in range(3):
                others = [d for d in range(3) if d != gd]
                acc = np.zeros((n, N_FEATS), np.float32)
                for k in range(4):
                    w = np.full(n, scale, np.float32)
                    cl = [None, None, None]
                    for b, d in enumerate(others):
                        if (k >> b) & 1:
                            w = (w * f[:, d]).astype(np.float32)
                            cl[d] = gi[:, d] + np.uint32(1)
                        else:
                            w = (w * (np.float32(1) - f[:, d])).astype(np.float32)
                            cl[d] = gi[:, d]
                    cl[gd] = gi[:, gd]
                    left = p16[_grid_index(meta, l, cl[0], cl[1], cl[2]) + off]
                    cl[gd] = gi[:, gd] + np.uint32(1)
                    right = p16[_grid_index(meta, l, cl[0], cl[1], cl[2]) + off]
                    diff = (right - left).astype(np.float32)
                    # grads += weight * diff  (one fused multiply-add)
                    acc = (w[:, None].astype(np.float64) * diff.astype(np.float64) + acc.astype(np.float64)).astype(np.float32)
                dydx[:, 2 * l:2 * l + 2, gd] = acc
    enc16 = torch.from_numpy(enc).to(torch.float16)
    if want_grad:
        return enc16, torch.from_numpy(dydx)
    return enc16


def _layer(h, W, accumulate):
    """h [n,k] @ W[out,k]^T on fp16-representable values -> fp32 tensor holding the layer's pre-activation as the chosen
    accumulator type would leave it."""
    if accumulate == "float":
        return h @ W.t()
    acc = torch.zeros(h.shape[0], W.shape[0], dtype=torch.float64)
    for k in range(0, h.shape[1], 16):     # one 16x16x16 matrix-unit step: exact products + sum, then the accumulator's rounding
        acc = (acc + h[:, k:k + 16].double() @ W[:, k:k + 16].double().t()).to(torch.float16).double()
    return acc.float()


def mlp_forward(x, params, n_in=67, n_out=3, width=64, accumulate=None):
    """tcnn FullyFusedMLP (2 hidden layers, ReLU, no bias) behind tcnn.Network: input cast to
    fp16 and padded to a multiple of 16 with ONES, weights fp16 row-major [out,in] in layer
    order, fp32 accumulation (`accumulate="half"`: half accumulator fragments, see ACCUMULATE), activations stored fp16,
    output sliced to n_out.  x [n,n_in]."""
    accumulate = accumulate or ACCUMULATE
    accumulate = "half" if accumulate == "half_mul_add" else accumulate
    pad_in = (n_in + 15) // 16 * 16
    pad_out = (n_out + 15) // 16 * 16
    w = params.detach().to(torch.float16).float()
    o = 0
    W1 = w[o:o + width * pad_in].view(width, pad_in); o += width * pad_in
    W2 = w[o:o + width * width].view(width, width); o += width * width
    W3 = w[o:o + pad_out * width].view(pad_out, width)
    n = x.shape[0]
    xin = torch.ones(n, pad_in)
    xin[:, :n_in] = x.detach().to(torch.float16).float()
    h = torch.relu(_layer(xin, W1, accumulate)).to(torch.float16).float()
    h = torch.relu(_layer(h, W2, accumulate)).to(torch.float16).float()
    out = _layer(h, W3, accumulate).to(torch.float16)
    return out[:, :n_out]


def mlp_num_params(n_in=67, n_out=3, width=64):
    return width * ((n_in + 15) // 16 * 16) + width * width + ((n_out + 15) // 16 * 16) * width


# ----------------------------------------------------------------------------------------
# Renderer.render_batch_ray sample placement (render.py:99-171)
# ----------------------------------------------------------------------------------------

def render_sample(rays_o, rays_d, gt_depth, bound, n_samples, n_surface, perturb_rand=None):
    """Returns z_vals, dists [n, n_samples + n_surface] exactly as render.py computes them
    (lindisp=False).  `perturb_rand` is the shared torch.rand(N_samples) vector (:159) or None."""
    n = rays_o.shape[0]
    if gt_depth is None:
        n_surface = 0
        near = 0.01
    else:
        gt_depth = gt_depth.reshape(-1, 1)
        near = gt_depth.repeat(1, n_samples) * 0.01
    t = (bound[None, :, :] - rays_o[:, :, None]) / rays_d[:, :, None]
    far_bb, _ = torch.min(torch.max(t, dim=2)[0], dim=1)
    far_bb = far_bb[:, None] + 0.01
    far = torch.clamp(far_bb, 0, (gt_depth * 1.2).max()) if gt_depth is not None else far_bb
    z_surf = None
    if n_surface > 0:
        valid = gt_depth > 0
        vd = (gt_depth * valid).repeat(1, n_surface)
        ts = torch.linspace(0, 1, steps=n_surface)[None, :].repeat(n, 1)
        snr, sfar = (1 - 0.1) * vd, (1 + 0.1) * vd
        zv = snr + (sfar - snr) * ts
        zi = 0.001 + (gt_depth.max() - 0.001) * ts
        z_surf = zv * valid + zi * (1 - valid.float())
    tv = torch.linspace(0, 1, steps=n_samples)[None, :].repeat(n, 1)
    z = near + (far - near) * tv
    sample_dist = ((far - near) / n_samples).mean(dim=1, keepdim=True)
    if perturb_rand is not None:
        mid = 0.5 * (z[:, :-1] + z[:, 1:])
        upper = torch.cat([mid, z[:, -1:]], 1)
        lower = torch.cat([z[:, :1], mid], 1)
        z = lower + (upper - lower) * perturb_rand
    if n_surface > 0:
        z, _ = torch.sort(torch.cat([z, z_surf.float()], 1), dim=1)
    dists = torch.cat([z[..., 1:] - z[..., :-1], sample_dist], -1)
    return z, dists


# ----------------------------------------------------------------------------------------
# InstantNeuS.forward (InstantNeuS.py:295-370)
# ----------------------------------------------------------------------------------------

def in_bound(pts, bound):
    m = (pts[:, 0] < bound[0, 1]) & (pts[:, 0] > bound[0, 0])
    m &= (pts[:, 1] < bound[1, 1]) & (pts[:, 1] > bound[1, 0])
    m &= (pts[:, 2] < bound[2, 1]) & (pts[:, 2] > bound[2, 0])
    return m


def sdf_and_gradient(pts, bound, grid, sdf_w, sdf_b, meta=None):
    """SDFNetwork.sdf(require_feature, require_gradient) (InstantNeuS.py:121-159): returns
    sdf [n,1], feat [n,31], gradient [n,3] = d sdf / d pts (analytic restatement of the autograd
    path: cat([x, enc]) -> Linear row 0; the encoding's input gradient is tcnn's dy_dx contracted
    with the fp16-cast upstream gradient)."""
    span = bound[:, 1] - bound[:, 0]
    p = (pts - bound[:, 0]) / span * 2.0 - 1.0
    inside = ((p >= -1.0) & (p <= 1.0)).float()
    p = p.clamp(-1.0, 1.0)
    view = (p + 1) / 2
    enc, dydx = grid_encode(view, grid, meta, want_grad=True)
    feat_in = torch.cat([p, enc.float()], dim=-1)
    out = feat_in @ sdf_w.t() + sdf_b
    g_enc = sdf_w[0, 3:].to(torch.float16).float()                  # dL/d enc arrives in fp16
    g_view = torch.einsum("ncd,c->nd", dydx, g_enc)
    g_p = sdf_w[0, :3][None] + g_view / 2
    grad = g_p * inside * 2.0 / span
    return out[:, :1], out[:, 1:], grad


def get_alpha(sdf, gradients, dirs, dists, inv_s):
    """InstantNeuS.py:276-293 with cos_anneal_ratio = 1."""
    true_cos = (dirs * gradients).sum(1, keepdim=True)
    iter_cos = -(torch.relu(-true_cos * 0.5 + 0.5) * 0.0 + torch.relu(-true_cos) * 1.0)
    d = dists.reshape(-1, 1)
    est_next = sdf + iter_cos * d / 2.0
    est_prev = sdf - iter_cos * d / 2.0
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    return ((prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)).clip(0.0, 1.0)


def neus_forward(rays_o, rays_d, z_vals, dists, P, meta=None):
    """P: dict(grid f32/f16 [total*2], sdf_w [32,35], sdf_b [32], color_B [3,33], mlp [10240],
    variance (python float), bound [3,2], rt_bound [3,2]).  Returns the reference's output dict."""
    n, s = z_vals.shape
    z_mid = z_vals + dists / 2.0
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * z_mid[:, :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(n, s, 3).reshape(-1, 3)
    mask = in_bound(pts, P["rt_bound"])
    if mask.float().sum() < 1:
        mask[:100] = True
    o_sdf, o_feat, o_grad = sdf_and_gradient(pts[mask], P["bound"], P["grid"], P["sdf_w"], P["sdf_b"], meta)
    npts = pts.shape[0]
    sdf = torch.ones(npts, 1) * 100
    grads = torch.zeros(npts, 3)
    feat = torch.zeros(npts, 31)
    sdf[mask], grads[mask], feat[mask] = o_sdf, o_grad, o_feat
    inv_s = float(min(max(math.exp(P["variance"] * 10.0), 1e-6), 1e6))
    alpha = get_alpha(sdf, grads, dirs, dists, inv_s)
    emb = torch.sin(pts[mask] @ P["color_B"])
    mlp_in = torch.cat([emb, grads[mask], feat[mask]], dim=1)
    o_rgb = torch.sigmoid(mlp_forward(mlp_in, P["mlp"]).float()).to(torch.float16)
    rgb = torch.zeros(npts, 3, dtype=torch.float16)
    rgb[mask] = o_rgb
    sdf = sdf.reshape(n, s)
    rgb = rgb.reshape(n, s, 3)
    alpha = (alpha * mask[:, None]).reshape(n, s)
    grads = grads.reshape(n, s, 3)
    m2 = mask.reshape(n, s)
    weights = alpha * torch.cumprod(torch.cat([torch.ones(n, 1), 1 - alpha + 1e-7], 1), 1)[:, :-1]
    weight_sum = weights.sum(1, keepdim=True)
    color = (rgb * weights[:, :, None]).sum(1)
    depth = (z_mid * weights).sum(1, keepdim=True)
    depth_var = ((z_mid - depth) ** 2 * weights).sum(1, keepdim=True)
    normals = ((grads * weights[:, :, None]) * m2[:, :, None]).sum(1)
    gerr = (torch.linalg.norm(grads, ord=2, dim=2) - 1.0) ** 2 * m2
    return {
        "color": color, "depth": depth, "depth_variance": depth_var, "normal": normals,
        "weight_sum": weight_sum, "sdf_variance": torch.full((n, 1), 1.0 / math.exp(P["variance"] * 10.0)),
        "sdf": sdf, "z_vals": z_mid, "gradient_error": gerr.mean().unsqueeze(0),
        # extras for kernel-level parity
        "_alpha": alpha, "_rgb": rgb, "_grad": grads, "_mask": m2,
    }


def compute_sdf_error(sdf, z_vals, gt_depth, truncation, sparse_factor):
    """InstantNeuS.py:372-400."""
    n, s = z_vals.shape
    pred = sdf.reshape(n, s)
    gt = gt_depth.reshape(n, 1)
    vm = (gt > 0).reshape(-1)
    gt, z, pred = gt[vm], z_vals[vm], pred[vm]
    front = z < (gt - truncation)
    bnd = gt - z
    sm = bnd.abs() <= truncation
    nvs = front.sum(1) + sm.sum(1) + 1e-8
    nvr = vm.sum()
    fl = torch.max(torch.exp((-sparse_factor * pred).clamp(max=10.0)) - torch.ones_like(pred), pred - bnd).clamp(min=0.0) * front
    sdf_front = (fl.sum(1) / nvs).sum() / nvr
    err = ((torch.abs(pred - bnd) * sm).sum(1) / nvs).sum() / nvr
    return err, sdf_front


def make_params(seed=0, grid_init=1e-4, bound=((-5.0, 5.0), (-5.0, 5.0), (-5.0, 5.0))):
    """Random-init parameter set with the reference's shapes/initialisers
    (InstantNeuS.py:108-112 sdf_layer init, :178 color _B, tcnn U(-1e-4,1e-4) grid,
    Xavier-uniform MLP); `grid_init=0.5` gives a 'trained-like' non-degenerate field."""
    g = torch.Generator().manual_seed(seed)
    meta = grid_meta()
    grid = (torch.rand(int(meta["total"]) * N_FEATS, generator=g) * 2 - 1) * grid_init
    if grid_init > 1e-3:   # trained-like: amplitude ~ 1/resolution so that |grad sdf| stays O(1)
        for l in range(N_LEVELS):
            a, b = 2 * int(meta["offset"][l]), 2 * (int(meta["offset"][l]) + int(meta["size"][l]))
            grid[a:b] *= 15.0 / float(meta["scale"][l])
    sdf_w = torch.zeros(32, 35)
    sdf_w[:, :3] = torch.randn(32, 3, generator=g) * (math.sqrt(2) / math.sqrt(32))
    if grid_init > 1e-3:   # trained-like: let the features matter
        sdf_w[:, 3:] = torch.randn(32, 32, generator=g) * 0.1
    sdf_b = torch.zeros(32)
    color_B = torch.randn(3, 33, generator=g) * 25.0

    def xavier(o, i):
        a = math.sqrt(6.0 / (o + i))
        return (torch.rand(o, i, generator=g) * 2 - 1) * a
    mlp = torch.cat([xavier(64, 80).reshape(-1), xavier(64, 64).reshape(-1), xavier(16, 64).reshape(-1)])
    b = torch.tensor(bound, dtype=torch.float32)
    return dict(grid=grid, sdf_w=sdf_w, sdf_b=sdf_b, color_B=color_B, mlp=mlp, variance=0.2,
                bound=b.clone(), rt_bound=b.clone())
