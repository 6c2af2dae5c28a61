#!/usr/bin/env python
"""Generate golden vectors by RUNNING THE REFERENCE'S OWN PYTHON CODE in the build container.

    python tests/golden/gen_golden.py            # needs /root/reference (absent on the GPU box)

The reference's native extensions cannot run here (CUDA-only `droid_backends`, `lietorch`
submodule empty, `tinycudann` not installed), so those imports are satisfied by stand-ins:

  droid_backends  -> thin wrappers over oracle/droid_oracle.py   (only so `import` succeeds and
                     CorrBlock.__call__'s level/permute/concat plumbing can execute)
  lietorch        -> go_slam_amd/lietorch_shim.py                (published SE3 conventions)
  tinycudann      -> autograd-capable wrappers over oracle/neus_oracle.py's HashGrid / MLP
  mcubes, trimesh -> empty modules (only used by mesh extraction)

What the fixtures therefore PIN is every line of reference Python on the hot path:
  * modules/corr.py      CorrBlock.corr + pyramid build (pure torch)          -> corr_*.npz
  * geom/projective_ops  projective_transform with and without Jacobians      -> proj_*.npz
  * render.py            Renderer.render_batch_ray sample placement           -> render_sample.npz, render_sample_mono.npz
  * nerf_func.py         build_rays                                            -> build_rays.npz
  * geom/ba.py + chol.py  BA: the pure-PyTorch dense bundle adjustment (one Gauss-Newton step)  -> ba_python.npz
  * droid_net.py + modules/gru.py  UpdateModule / ConvGRU / GraphAgg / cvx_upsample (CPU fp32)  -> update_module.npz
  * factor_graph.py      edge management: duplicate filter, max_factors retirement, filter_edges,
                         rm_keyframe, neighbourhood / proximity proposals with NMS  -> factor_graph_edges.npz
  * backend.py           Backend.ba edge proposal (dense + loop closure)       -> backend_edges.npz
  * frontend.py          Frontend: call sequence into the graph / loop closure  -> frontend_trace.npz
  * modules/extractor.py BasicEncoder (fnet / cnet) + DroidNet checkpoint keys   -> encoders.npz
  * motion_filter.py     MotionFilter.track keyframe decisions + appended items  -> motion_filter.npz
  * multiview_filter.py  MultiviewFilter.forward host logic (masks, bound, priority) -> multiview_filter.npz
  * trajectory_filler.py PoseTrajectoryFiller: bracketing, interpolation, parked items, edges -> trajectory_filler.npz
  * mapping.py + depth_video.get_mapping_item  Mapper keyframe schedule and ray batches -> mapper.npz
  * mapping.py           Mapper.optimize_map: the loss and its gradients w.r.t. the renderer outputs -> mapper_loss.npz
  * factor_graph.py      update_lowmem chunking and call arguments (mocked kernels) -> update_lowmem.npz
  * InstantNeuS.py       normalisation, masking, sdf gradient by autograd.grad, get_alpha,
                         compositing, compute_sdf_error                       -> neus_forward.npz, neus_forward_cases.npz
  * InstantNeuS.py       the SAME forward differentiated by its own autograd graph (autograd.grad(create_graph=True) +
                         backward) on a twice-differentiable stand-in: every trained parameter's gradient -> neus_backward.npz
  * lib/*.cu             the reference's CUDA kernels themselves, compiled for the CPU (oracle/build_ref.py): ba,
                         frame_distance, projmap, iproj, depth_filter on seeded inputs             -> reference_kernels.npz
  * modules/corr.py + lib/altcorr_kernel.cu  AltCorrBlock over the reference's own alt-corr kernel, 4 levels, 9 edges
                                                                                -> reference_altcorr_pyramid.npz
The tcnn / lietorch stand-ins stay "parity unpinned" (DESIGN.md 5); the CUDA kernels no longer do.
The reference is only imported, never copied; outputs are small .npz files next to this script.
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from go_slam_amd import lietorch_shim, synth  # noqa: E402
from oracle import droid_oracle as DO, neus_oracle as NO  # noqa: E402


def install_stubs():
    # --- droid_backends
    db = types.ModuleType("droid_backends")
    db.corr_index_forward = lambda vol, coords, r: DO.corr_index_forward(vol, coords, r)
    db.corr_index_backward = lambda vol, coords, g, r: DO.corr_index_backward(vol, coords, g, r)
    db.altcorr_forward = lambda f1, f2, c, r: DO.altcorr_forward(f1, f2, c, r)
    db.iproj = lambda poses, disps, intr: DO.iproj(poses, disps, intr)
    db.depth_filter = lambda poses, disps, intr, ix, th: DO.depth_filter(poses, disps, intr, ix, th)
    if "colorama" not in sys.modules:
        try:
            importlib.import_module("colorama")
        except ImportError:
            col = types.ModuleType("colorama")
            class _Blank:
                def __getattr__(self, name):
                    return ""
            col.Fore = col.Style = _Blank()
            sys.modules["colorama"] = col
    sys.modules["droid_backends"] = db
    # --- lietorch
    lt = types.ModuleType("lietorch")
    lt.SE3, lt.Sim3, lt.cat = lietorch_shim.SE3, lietorch_shim.Sim3, lietorch_shim.cat
    sys.modules["lietorch"] = lt
    # --- tinycudann
    tc = types.ModuleType("tinycudann")

    class _GridFn(torch.autograd.Function):
        @staticmethod
        def forward(ctx, x, params):
            enc, dydx = NO.grid_encode(x, params, want_grad=True)
            ctx.save_for_backward(dydx)
            return enc

        @staticmethod
        def backward(ctx, g):
            dydx, = ctx.saved_tensors
            # tcnn: dL/dx = sum_c float(dL/dy_c as fp16) * dy_dx
            return torch.einsum("nc,ncd->nd", g.to(torch.float16).float(), dydx), None

    class Encoding(torch.nn.Module):
        def __init__(self, n_input_dims, encoding_config, **kw):
            super().__init__()
            self.n_input_dims, self.n_output_dims = n_input_dims, 32
            self.params = torch.nn.Parameter(torch.zeros(int(NO.grid_meta()["total"]) * 2))

        def forward(self, x):
            return _GridFn.apply(x, self.params)

    class Network(torch.nn.Module):
        def __init__(self, n_input_dims, n_output_dims, network_config, **kw):
            super().__init__()
            self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
            self.params = torch.nn.Parameter(torch.zeros(NO.mlp_num_params(n_input_dims, n_output_dims)))

        def forward(self, x):
            return NO.mlp_forward(x, self.params, self.n_input_dims, self.n_output_dims)

    tc.Encoding, tc.Network = Encoding, Network
    sys.modules["tinycudann"] = tc
    for name in ("mcubes", "trimesh"):
        sys.modules[name] = types.ModuleType(name)
    # the reference wraps tcnn construction in `with torch.cuda.device(device)`; no GPU here
    class _NoDev:
        def __init__(self, *a, **k): pass
        def __enter__(self): return self
        def __exit__(self, *a): return False
    torch.cuda.device = _NoDev
    # synthetic package so relative imports work without executing src/__init__.py
    pkg = types.ModuleType("refsrc")
    pkg.__path__ = [os.path.join(REF, "src")]
    sys.modules["refsrc"] = pkg
    for sub in ("modules", "geom"):
        sp = types.ModuleType(f"refsrc.{sub}")
        sp.__path__ = [os.path.join(REF, "src", sub)]
        sys.modules[f"refsrc.{sub}"] = sp


def save(name, **arrays):
    out = {k: (v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in arrays.items()}
    np.savez_compressed(os.path.join(HERE, name), **out)
    print("wrote", name, {k: v.shape for k, v in out.items()})


def gen_corr():
    corr = importlib.import_module("refsrc.modules.corr")
    g = torch.Generator().manual_seed(101)
    for tag, dt in (("f32", torch.float32), ("f16", torch.float16)):
        f1 = torch.randn(1, 2, 128, 16, 16, generator=g).to(dt)
        f2 = torch.randn(1, 2, 128, 16, 16, generator=g).to(dt)
        blk = corr.CorrBlock(f1, f2)
        coords = torch.stack(torch.meshgrid(torch.arange(16.0), torch.arange(16.0), indexing="xy"), -1)
        coords = coords[None, None].repeat(1, 2, 1, 1, 1) + 2.5 * torch.randn(1, 2, 16, 16, 2, generator=g)
        out = blk(coords)
        save(f"corr_{tag}.npz", fmap1=f1.float(), fmap2=f2.float(), coords=coords, lookup=out.float(),
             **{f"pyr{i}": p.float() for i, p in enumerate(blk.corr_pyramid)})


def gen_proj():
    pops = importlib.import_module("refsrc.geom.projective_ops")
    vid = synth.make_video(7, "tiny", seed=103)
    ii, jj = synth.make_graph(7, 18, seed=103)
    ii = torch.cat([ii, torch.tensor([3])])
    jj = torch.cat([jj, torch.tensor([3])])            # one stereo edge
    Gs = lietorch_shim.SE3(vid["poses"][None])
    coords, valid = pops.projective_transform(Gs, vid["disps"][None], vid["intrinsics"][None], ii, jj)
    c2, v2, (Ji, Jj, Jz) = pops.projective_transform(Gs, vid["disps"][None], vid["intrinsics"][None], ii, jj,
                                                     jacobian=True)
    save("proj.npz", poses=vid["poses"], disps=vid["disps"], intrinsics=vid["intrinsics"], ii=ii, jj=jj,
         coords=coords, valid=valid, Ji=Ji, Jj=Jj, Jz=Jz)


def gen_render():
    render = importlib.import_module("refsrc.render")
    cfg = {"rendering": {"lindisp": False, "perturb": 1.0, "N_samples": 24, "N_surface": 48}}
    slam = types.SimpleNamespace(H=480, W=640, fx=577.0, fy=578.0, cx=319.0, cy=242.0)
    R = render.Renderer(cfg, None, slam)
    g = torch.Generator().manual_seed(107)
    n = 97
    o = torch.rand(n, 3, generator=g) * 4 - 2
    d = torch.randn(n, 3, generator=g)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[::9] = 0
    bound = torch.tensor([[-5.0, 5.0], [-4.0, 4.5], [-3.0, 6.0]])
    grabbed = {}

    class Net:
        def __init__(self): self.bound = bound
        def __call__(self, ro, rd, zv, ds, render_params=None):
            grabbed["z"], grabbed["d"] = zv.clone(), ds.clone()
            return {"z": zv}
    torch.manual_seed(1234)
    R.render_batch_ray(o, d, Net(), None, device="cpu", gt_depth=gt)
    torch.manual_seed(1234)
    pr = torch.rand(24)
    z1, d1 = grabbed["z"], grabbed["d"]
    torch.manual_seed(1234)
    R.render_batch_ray(o, d, Net(), None, device="cpu", gt_depth=None)
    save("render_sample.npz", rays_o=o, rays_d=d, gt_depth=gt, bound=bound, perturb=pr, z_depth=z1, dists_depth=d1,
         z_nodepth=grabbed["z"], dists_nodepth=grabbed["d"])


def gen_render_mono():
    """configs[4] (configs/Replica/replica_mono.yaml:54-55): 48 stratified + 24 near-surface samples, and the degenerate
    rays a batch can contain: no depth measurement, the box exit behind the camera (far < near: the stratified run comes out
    DESCENDING before the reference's sort), an origin outside the bound, a batch whose depth maximum is tiny."""
    render = importlib.import_module("refsrc.render")
    cfg = {"rendering": {"lindisp": False, "perturb": 1.0, "N_samples": 48, "N_surface": 24}}
    slam = types.SimpleNamespace(H=340, W=600, fx=300.0, fy=300.0, cx=299.5, cy=169.5)
    R = render.Renderer(cfg, None, slam)
    g = torch.Generator().manual_seed(109)
    n = 131
    o = torch.rand(n, 3, generator=g) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[::7] = 0
    bound = torch.tensor([[-2.5, 2.5], [-2.0, 2.25], [-1.5, 3.0]])
    o[3] = torch.tensor([4.0, 0.1, 0.2]); d[3] = torch.tensor([1.0, 0.0, 0.0])       # outside, looking away: exit behind
    o[4] = torch.tensor([-3.5, 0.0, 0.5]); d[4] = torch.tensor([-0.6, 0.8, 0.0])
    o[5] = torch.tensor([2.4, 2.2, 2.9]); d[5] = torch.nn.functional.normalize(torch.tensor([1.0, 1.0, 1.0]), dim=0)
    gt[3], gt[4], gt[5] = 1.5, 0.0, 2.0
    grabbed = {}

    class Net:
        def __init__(self): self.bound = bound
        def __call__(self, ro, rd, zv, ds, render_params=None):
            grabbed["z"], grabbed["d"] = zv.clone(), ds.clone()
            return {"z": zv}
    out = {}
    for tag, depth in (("depth", gt), ("tiny", gt * 2e-4), ("nodepth", None)):
        torch.manual_seed(4321)
        R.render_batch_ray(o, d, Net(), None, device="cpu", gt_depth=depth)
        out["z_" + tag], out["dists_" + tag] = grabbed["z"], grabbed["d"]
    torch.manual_seed(4321)
    pr = torch.rand(48)
    save("render_sample_mono.npz", rays_o=o, rays_d=d, gt_depth=gt, bound=bound, perturb=pr, **out)


def gen_neus():
    neus = importlib.import_module("refsrc.InstantNeuS")
    P = NO.make_params(109, grid_init=0.3, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
    rt = torch.tensor([[-2.2, 2.3], [-2.4, 2.1], [-2.0, 2.2]])
    cfg = {"sdf_network": {"d_in": 3, "d_out": 32}, "color_network": {"d_in": 3, "d_feat": 31, "d_hidden": 64, "n_layers": 2},
           "variance_network": {"init_val": 0.2, "scale_factor": 10.0}, "sdf_smooth_std": 0.005,
           "sdf_sparse_factor": 5, "sdf_truncation": 0.16, "sdf_random_weight": 0.04}
    torch.manual_seed(5)
    net = neus.InstantNeuS(cfg, P["bound"].tolist(), device="cpu")
    with torch.no_grad():
        net.sdf_network.encoding.encoding.params.copy_(P["grid"])
        net.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
        net.sdf_network.sdf_layer.bias.copy_(P["sdf_b"])
        net.color_network._B.copy_(P["color_B"])
        net.color_network.network.params.copy_(P["mlp"])
    net.update_bound(rt)
    g = torch.Generator().manual_seed(113)
    n = 40
    o = torch.rand(n, 3, generator=g) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[::7] = 0
    z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, torch.rand(24, generator=g))
    out = net(o, d, z, dist)
    sdf_err, sdf_front = net.compute_sdf_error(out["sdf"], out["z_vals"], gt)
    sd = net.state_dict()
    names = {id(p): k for k, p in net.named_parameters()}
    train_names = [names[id(p)] for p in net.get_training_parameters()]
    volume_names = [names[id(p)] for p in net.get_volume_parameters()]
    save("neus_forward.npz", seed=109, rt_bound=rt, rays_o=o, rays_d=d, gt_depth=gt, z_in=z, dists_in=dist,
         sdf_error=sdf_err, sdf_front_error=sdf_front, **{k: v for k, v in out.items()},
         state_keys=np.array(list(sd.keys())), state_shapes=np.array([str(tuple(v.shape)) for v in sd.values()]),
         train_param_names=np.array(train_names), volume_param_names=np.array(volume_names))
    # the degenerate batches of InstantNeuS.py:309-312: (a) NO point inside the realtime bound -> the first 100 points are
    # forced valid; (b) a realtime bound that cuts through the rays (most samples masked out: sdf 100, alpha 0); (c) points
    # outside the STATIC bound but inside the realtime one (normalisation clamps them, `inside` zeroes their gradient)
    cases = {"forced": torch.tensor([[50.0, 51.0], [50.0, 51.0], [50.0, 51.0]]),
             "cut": torch.tensor([[-0.4, 0.5], [-2.4, 2.1], [-0.3, 0.6]]),
             "wide": torch.tensor([[-4.0, 4.0], [-4.0, 4.0], [-4.0, 4.0]])}
    extra = {}
    for tag, rtb in cases.items():
        net.update_bound(rtb)
        oc = net(o, d, z, dist)
        extra["rt_" + tag] = rtb
        for k, v in oc.items():
            extra[f"{k}_{tag}"] = v
    save("neus_forward_cases.npz", seed=109, rays_o=o, rays_d=d, gt_depth=gt, z_in=z, dists_in=dist, **extra)


def differentiable_tcnn():
    """Swap the tinycudann stand-in for a TWICE-differentiable one (pure torch ops from oracle/neus_autograd.py: the
    hash-grid interpolation as a function of its input AND its table, the MLP with straight-through fp16 roundings), so
    that the reference module's own `autograd.grad(sdf, pts, create_graph=True)` + `backward()` produce gradients of
    every trained parameter.  The gradient arriving at the encoding output is rounded to fp16 as tcnn does
    (kernel_grid_backward_input reads dL/dy as __half), with an identity second derivative."""
    from oracle import neus_autograd as NA
    tc = sys.modules["tinycudann"]
    meta = NO.grid_meta()

    class _HalfGrad(torch.autograd.Function):
        @staticmethod
        def forward(ctx, y):
            return y.clone()

        @staticmethod
        def backward(ctx, g):
            return g + (g.to(torch.float16).to(g.dtype) - g).detach()

    class Encoding(torch.nn.Module):
        def __init__(self, n_input_dims, encoding_config, **kw):
            super().__init__()
            self.n_input_dims, self.n_output_dims = n_input_dims, 32
            self.params = torch.nn.Parameter(torch.zeros(int(meta["total"]) * 2))

        def forward(self, x):
            enc, _ = NA.grid_encode_diff(x, self.params, meta, x_differentiable=True)
            return _HalfGrad.apply(enc)

    class Network(torch.nn.Module):
        def __init__(self, n_input_dims, n_output_dims, network_config, **kw):
            super().__init__()
            self.n_input_dims, self.n_output_dims = n_input_dims, n_output_dims
            self.params = torch.nn.Parameter(torch.zeros(NO.mlp_num_params(n_input_dims, n_output_dims)))

        def forward(self, x):
            return NA.mlp_diff(x, self.params, self.n_input_dims, self.n_output_dims)
    keep = (tc.Encoding, tc.Network)
    tc.Encoding, tc.Network = Encoding, Network
    return keep


def gen_neus_backward():
    """The reference's `InstantNeuS.forward` (src/InstantNeuS.py:295-370: masking, normalisation, the sdf gradient by
    autograd.grad(create_graph=True), get_alpha, sin embedding, compositing) DIFFERENTIATED BY ITS OWN AUTOGRAD GRAPH on
    the twice-differentiable tcnn stand-in, under the mapper's loss (src/mapping.py:96-132; pinned separately by
    mapper_loss.npz) -> the gradient of every trained parameter.  Pins oracle/neus_autograd.py's explicit formulation
    (analytic sdf gradient, second-order terms) -- the referee of the HIP training backward."""
    from oracle import neus_autograd as NA
    keep = differentiable_tcnn()
    tc = sys.modules["tinycudann"]
    try:
        neus = importlib.reload(importlib.import_module("refsrc.InstantNeuS"))
        seed = 127
        P = NO.make_params(seed, grid_init=0.3, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
        rt = torch.tensor([[-2.2, 2.3], [-2.4, 2.1], [-2.0, 2.2]])
        cfg = {"sdf_network": {"d_in": 3, "d_out": 32},
               "color_network": {"d_in": 3, "d_feat": 31, "d_hidden": 64, "n_layers": 2},
               "variance_network": {"init_val": 0.2, "scale_factor": 10.0}, "sdf_smooth_std": 0.005,
               "sdf_sparse_factor": 5, "sdf_truncation": 0.16, "sdf_random_weight": 0.04}
        torch.manual_seed(5)
        net = neus.InstantNeuS(cfg, P["bound"].tolist(), device="cpu")
        with torch.no_grad():
            net.sdf_network.encoding.encoding.params.copy_(P["grid"])
            net.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
            net.sdf_network.sdf_layer.bias.copy_(P["sdf_b"])
            net.color_network._B.copy_(P["color_B"])
            net.color_network.network.params.copy_(P["mlp"])
        net.update_bound(rt)
        g = torch.Generator().manual_seed(131)
        n = 12
        o = torch.rand(n, 3, generator=g) * 4 - 2
        d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
        gt = torch.rand(n, generator=g) * 3.5 + 0.5
        gt[::5] = 0
        col = torch.rand(n, 3, generator=g)
        z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, torch.rand(24, generator=g))
        with torch.enable_grad():
            out = net(o, d, z, dist)
            loss = NA.mapping_loss(out, col, gt)
            loss.backward()
        grid_g = net.sdf_network.encoding.encoding.params.grad
        nz = torch.nonzero(grid_g).reshape(-1)
        save("neus_backward.npz", seed=seed, rt_bound=rt, rays_o=o, rays_d=d, gt_depth=gt, rays_color=col, z_in=z,
             dists_in=dist, loss=loss.detach(), color=out["color"].detach(), depth=out["depth"].detach(),
             sdf=out["sdf"].detach(), gradient_error=out["gradient_error"].detach(),
             g_sdf_w=net.sdf_network.sdf_layer.weight.grad, g_sdf_b=net.sdf_network.sdf_layer.bias.grad,
             g_color_B=net.color_network._B.grad, g_mlp=net.color_network.network.params.grad,
             g_variance=net.variance_network.variance.grad.reshape(1), g_grid_index=nz, g_grid_value=grid_g[nz])
    finally:
        tc.Encoding, tc.Network = keep
        importlib.reload(importlib.import_module("refsrc.InstantNeuS"))


def gen_rays():
    nf = importlib.import_module("refsrc.nerf_func")
    g = torch.Generator().manual_seed(127)
    H, W = 24, 32
    depth = torch.rand(H, W, generator=g) * 3 + 0.5
    color = torch.rand(H, W, 3, generator=g)
    mask = torch.rand(H, W, generator=g) > 0.3
    c2w = torch.eye(4)
    c2w[:3, :3] = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
    c2w[:3, 3] = torch.tensor([0.3, -0.2, 1.0])
    torch.manual_seed(77)
    o, d, dep, col = nf.build_rays(2, 22, 3, 29, 50, H, W, 30.0, 31.0, 15.5, 11.5, c2w, depth, color, "cpu", mask=mask)
    save("build_rays.npz", depth=depth, color=color, mask=mask, c2w=c2w, rays_o=o, rays_d=d, ray_depth=dep, ray_color=col)


def named_weights(state_dict, seed=131):
    """Deterministic weights by parameter NAME (so the test can rebuild them without storing 10 MB):
    randn from a generator seeded with the name's crc32, scaled like a fan-in init."""
    import zlib
    out = {}
    for name, t in state_dict.items():
        g = torch.Generator().manual_seed(seed + zlib.crc32(name.encode()) % (2 ** 31))
        fan_in = t[0].numel() if t.dim() > 1 else 16
        out[name] = torch.randn(t.shape, generator=g) / (fan_in ** 0.5)
    return out


def gen_update():
    """The reference's own UpdateModule / ConvGRU / GraphAgg (src/droid_net.py:34-140, src/modules/gru.py) on a
    tiny graph, CPU fp32.  torch_scatter is absent: scatter_mean is stood in by its published definition."""
    ts = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=1):
        assert dim == 1
        n = int(index.max()) + 1
        out = torch.zeros(src.shape[0], n, *src.shape[2:], dtype=src.dtype)
        out.index_add_(1, index, src)
        cnt = torch.bincount(index, minlength=n).clamp(min=1).to(src.dtype)
        return out / cnt.view(1, -1, *([1] * (src.dim() - 2)))
    ts.scatter_mean = scatter_mean
    sys.modules["torch_scatter"] = ts
    mods = sys.modules["refsrc.modules"]
    mods.GradientClip = importlib.import_module("refsrc.modules.clipping").GradientClip
    mods.ConvGRU = importlib.import_module("refsrc.modules.gru").ConvGRU
    mods.BasicEncoder = importlib.import_module("refsrc.modules.extractor").BasicEncoder
    corr = importlib.import_module("refsrc.modules.corr")
    mods.CorrBlock, mods.AltCorrBlock = corr.CorrBlock, corr.AltCorrBlock
    dn = importlib.import_module("refsrc.droid_net")
    op = dn.UpdateModule().eval()
    op.load_state_dict(named_weights(op.state_dict()))
    g = torch.Generator().manual_seed(137)
    E, h, w = 5, 6, 8
    net = torch.tanh(torch.randn(1, E, 128, h, w, generator=g))
    inp = torch.relu(torch.randn(1, E, 128, h, w, generator=g))
    corr_f = 0.5 * torch.randn(1, E, 196, h, w, generator=g)
    flow = torch.randn(1, E, 4, h, w, generator=g)
    ii = torch.tensor([2, 0, 2, 1, 0])
    jj = torch.tensor([0, 1, 1, 2, 2])
    n2, delta, weight, eta, upmask = op(net, inp, corr_f, flow, ii, jj)
    data = torch.rand(3, h, w, 1, generator=g)
    up = dn.cvx_upsample(data, upmask[0])
    save("update_module.npz", net=net, inp=inp, corr=corr_f, flow=flow, ii=ii, jj=jj, net_out=n2, delta=delta,
         weight=weight, eta=eta, upmask=upmask, up_data=data, up_out=up,
         keys=np.array(sorted(op.state_dict().keys())))


def gen_ba():
    """The reference's pure-PyTorch dense bundle adjustment (src/geom/ba.py BA + src/geom/chol.py schur_solve; the
    formulation its CUDA `ba` kernel was derived from): one Gauss-Newton step on a small monocular graph in
    which every keyframe is the source of at least one edge (so both paths optimise the same depth maps)."""
    ts = sys.modules.get("torch_scatter") or types.ModuleType("torch_scatter")

    def scatter_sum(src, index, dim=1, dim_size=None):
        assert dim == 1
        out = torch.zeros(src.shape[0], dim_size, *src.shape[2:], dtype=src.dtype)
        return out.index_add_(1, index, src)
    ts.scatter_sum = scatter_sum
    if not hasattr(ts, "scatter_mean"):
        ts.scatter_mean = None
    sys.modules["torch_scatter"] = ts
    sys.modules["projective_ops"] = importlib.import_module("refsrc.geom.projective_ops")   # `import projective_ops`
    ba = importlib.import_module("refsrc.geom.ba")
    N, shape = 6, "tiny"
    p = synth.make_ba_problem(N, 16, shape, seed=139, rgbd=False)
    ii = torch.cat([p["ii"], torch.arange(N)])                 # every keyframe is a source at least once
    jj = torch.cat([p["jj"], (torch.arange(N) + 1) % N])
    c, _ = DO.reproject(p["poses"], p["disps"], p["intrinsics"], ii, jj)
    g = torch.Generator().manual_seed(149)
    E = len(ii)
    ht, wd = p["disps"].shape[-2:]
    target = c[0] + 0.5 * torch.randn(E, ht, wd, 2, generator=g)
    weight = torch.rand(E, ht, wd, 2, generator=g)
    eta = 1e-2 * torch.rand(N, ht, wd, generator=g) + 1e-4
    Gs = lietorch_shim.SE3(p["poses"][None].clone())
    poses_out, disps_out = ba.BA(target[None], weight[None], eta[None], Gs, p["disps"][None].clone(),
                                 p["intrinsics"][None], ii, jj, fixedp=1)
    save("ba_python.npz", poses=p["poses"], disps=p["disps"], intrinsics=p["intrinsics"], ii=ii, jj=jj,
         target=target, weight=weight, eta=eta, poses_out=poses_out.data[0], disps_out=disps_out[0])


def graph_script(t_total=12, seed=151):
    """The deterministic edge-management scenario both sides replay: (distance matrix, per-step weights)."""
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(t_total, t_total, generator=g) * 30.0
    dist = 0.5 * (a + a.T) + 3.0 * (torch.arange(t_total)[:, None] - torch.arange(t_total)[None]).abs().float()
    conf = torch.rand(256, generator=g)                         # per-edge confidences for filter_edges
    conf = torch.where(conf < 0.35, conf * 1e-3, conf)
    return dist, conf


def run_graph_scenario(graph, video, dist, conf, set_counter, snapshot):
    """Drives a FactorGraph (the reference's or ours -- same method names) through neighbourhood / proximity
    proposals, the max_factors retirement, confidence filtering, keyframe removal, ageing and clear_edges,
    calling snapshot(tag) after every step."""
    def set_weight():
        E = graph.ii.shape[0]
        graph.weight = conf[:E].view(1, E, 1, 1, 1).expand(1, E, graph.ht, graph.wd, 2).clone()
    set_counter(6)
    graph.add_neighborhood_factors(0, 6, r=2); snapshot("nbr")
    graph.age += 1
    set_counter(9)
    graph.add_proximity_factors(t0=3, t1=0, rad=2, nms=2, thresh=40.0, remove=True); snapshot("prox1")
    graph.age += 1
    set_weight()
    graph.filter_edges(); snapshot("filter")
    graph.rm_keyframe(7); snapshot("rmkf")
    set_counter(12)
    graph.add_proximity_factors(t0=7, t1=2, rad=2, nms=2, thresh=45.0, remove=True); snapshot("prox2")
    graph.age += 1
    graph.rm_factors(graph.age > 2, store=True); snapshot("age")
    graph.add_proximity_factors(t0=0, t1=0, rad=1, nms=1, thresh=30.0, remove=True); snapshot("prox3")
    graph.add_proximity_factors(t0=5, t1=1, rad=3, nms=0, beta=0.5, thresh=60.0, remove=False, max_t=10)
    snapshot("prox4")
    graph.clear_edges(); snapshot("clear")
    graph.add_neighborhood_factors(8, 12, r=3); snapshot("nbr2")


def gen_graph():
    """The reference's own FactorGraph edge management (src/factor_graph.py:43-197, 368-450) on CPU with a mocked
    video (frame distances come from a fixed matrix, reprojection returns zeros); records every edge list after
    every step of run_graph_scenario."""
    import contextlib
    mods = sys.modules["refsrc.modules"]
    corr = importlib.import_module("refsrc.modules.corr")
    mods.CorrBlock, mods.AltCorrBlock = corr.CorrBlock, corr.AltCorrBlock
    fg = importlib.import_module("refsrc.factor_graph")
    T, h, w = 12, 16, 16
    dist, conf = graph_script(T)
    gen = torch.Generator().manual_seed(157)
    B = T + 2
    video = types.SimpleNamespace(
        ht=8 * h, wd=8 * w, stereo=False, counter=types.SimpleNamespace(value=0),
        get_lock=contextlib.nullcontext,
        distance=lambda ii, jj, beta=0.3: dist[ii, jj].clone(),
        reproject=lambda ii, jj: (torch.zeros(1, len(ii), h, w, 2), torch.ones(1, len(ii), h, w, 1)))
    for name, shape in (("timestamp", ()), ("images", (3, 8 * h, 8 * w)), ("dirty", ()), ("red", ()), ("poses", (7,)),
                        ("poses_gt", (4, 4)), ("disps", (h, w)), ("disps_sens", (h, w)), ("disps_up", (8 * h, 8 * w)),
                        ("depths_gt", (8 * h, 8 * w)), ("intrinsics", (4,)), ("poses_filtered", (7,)),
                        ("disps_filtered", (h, w)), ("mask_filtered", (h, w)), ("update_priority", ()),
                        ("nets", (4, h, w)), ("inps", (4, h, w)), ("fmaps", (1, 4, h, w))):
        setattr(video, name, torch.rand(B, *shape, generator=gen))
    poses0 = video.poses.clone()
    graph = fg.FactorGraph(video, None, device="cpu", corr_impl="volume", max_factors=40)
    out = {"dist": dist, "conf": conf, "poses_in": poses0}
    tags = []

    def snapshot(tag):
        tags.append(tag)
        for k in ("ii", "jj", "age", "ii_inac", "jj_inac", "ii_bad", "jj_bad"):
            out[f"{tag}_{k}"] = getattr(graph, k).clone()
    run_graph_scenario(graph, video, dist, conf, lambda n: setattr(video.counter, "value", n), snapshot)
    out["poses_out"] = video.poses
    save("factor_graph_edges.npz", tags=np.array(tags), **out)


BACKEND_CASES = (  # name, t_start, t_end, nms, radius, thresh, max_factors, t_start_loop, loop, stereo
    ("dense", 0, 14, 2, 2, 28.0, 112, None, False, False),
    ("dense_off", 3, 17, 1, 1, 35.0, 84, None, False, False),
    ("stereo", 0, 10, 1, 2, 30.0, 90, None, False, True),
    ("capped", 0, 16, 0, 1, 50.0, 40, None, False, False),
    ("loop", 0, 20, 2, 3, 34.0, 64, 12, True, False),
    ("loop_stereo", 2, 18, 1, 2, 40.0, 48, 9, True, True),
    ("few", 0, 2, 2, 2, 1.0, 16, None, False, False),
)


class RecorderGraph:
    """stands in for FactorGraph in Backend.ba: records what the backend asks of it."""

    def __init__(self):
        self.ii = torch.zeros(0, dtype=torch.long)
        self.calls = []

    def add_factors(self, ii, jj, remove=False):
        self.calls.append(("add_factors", bool(remove)))
        self.es = torch.stack([torch.as_tensor(ii), torch.as_tensor(jj)], 1).long()
        self.ii = self.es[:, 0]

    def update_lowmem(self, **kw):
        self.calls.append(("update_lowmem", tuple(sorted(kw.items()))))

    def clear_edges(self):
        self.calls.append(("clear_edges",))


def backend_distance(t_total=20, seed=163):
    g = torch.Generator().manual_seed(seed)
    a = torch.rand(t_total, t_total, generator=g) * 40.0
    a = 0.5 * (a + a.T)
    # a band of revisits so the loop detector sees 3x3 neighbourhoods below threshold
    for k in range(3):
        for o in (-1, 0, 1):
            for p in (-1, 0, 1):
                a[15 + k + o, 2 + k + p] = a[2 + k + p, 15 + k + o] = 6.0 + k + 0.1 * o + 0.01 * p
    return a


def backend_cfg():
    return {"tracking": {"upsample": False, "beta": 0.75, "backend": {
        "thresh": 25.0, "radius": 1, "nms": 5, "loop_window": 25, "loop_thresh": 25.0, "loop_radius": 1,
        "loop_nms": 12}}}


def run_backend_cases(backend_cls, dist):
    """Replays BACKEND_CASES through a Backend class (the reference's or ours)."""
    out = {}
    for name, ts, te, nms, rad, th, mf, tsl, loop, stereo in BACKEND_CASES:
        video = types.SimpleNamespace(stereo=stereo, dirty=torch.zeros(dist.shape[0], dtype=torch.bool),
                                      distance=lambda ii, jj, beta=0.3: dist[ii, jj].clone())
        be = backend_cls(types.SimpleNamespace(update=None), video, types.SimpleNamespace(device="cpu"), backend_cfg())
        graph = RecorderGraph()
        n = be.ba(ts, te, 4, graph, nms, rad, th, mf, t_start_loop=tsl, loop=loop, motion_only=(name == "loop"))
        out[name] = {"ret": n, "es": getattr(graph, "es", torch.zeros(0, 2, dtype=torch.long)), "calls": graph.calls,
                     "dirty": video.dirty.clone()}
    return out


def gen_backend():
    """The reference's Backend.ba edge proposal (src/backend.py:25-120), dense and loop-closure mode, on CPU with a
    fixed distance matrix and a recording graph."""
    importlib.import_module("refsrc.factor_graph")
    be = importlib.import_module("refsrc.backend")
    dist = backend_distance()
    res = run_backend_cases(be.Backend, dist)
    arrays = {"dist": dist}
    for name, r in res.items():
        arrays[f"{name}_es"] = r["es"]
        arrays[f"{name}_ret"] = np.array(r["ret"])
        arrays[f"{name}_dirty"] = r["dirty"]
        arrays[f"{name}_calls"] = np.array(repr(r["calls"]))
    save("backend_edges.npz", **arrays)


class TraceGraph:
    """stands in for FactorGraph under Frontend: keeps just enough state (ii, age, corr) for the frontend's
    control flow and records every call with its arguments."""
    trace = None

    def __init__(self, video, update_op, device="cpu", corr_impl="volume", max_factors=-1, upsample=False):
        self.video = video
        self.ii = torch.zeros(0, dtype=torch.long)
        self.jj = torch.zeros(0, dtype=torch.long)
        self.age = torch.zeros(0, dtype=torch.long)
        self.corr = None
        TraceGraph.trace.append(("graph", corr_impl, max_factors, bool(upsample)))

    def _add(self, lo, hi):
        new = torch.arange(lo, hi)
        self.ii = torch.cat([self.ii, new])
        self.jj = torch.cat([self.jj, new])
        self.age = torch.cat([self.age, torch.zeros_like(new)])
        self.corr = "volumes"

    def add_neighborhood_factors(self, t0, t1, r=3):
        TraceGraph.trace.append(("nbr", t0, t1, r))
        self._add(t0, t1)

    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False, max_t=None):
        cnt = self.video.counter
        TraceGraph.trace.append(("prox", t0, t1, rad, nms, beta, thresh, bool(remove), max_t))
        self._add(max(t0, 0), int(getattr(cnt, "value", cnt)))

    def rm_factors(self, mask, store=False):
        TraceGraph.trace.append(("rm", mask.tolist(), bool(store)))
        self.ii, self.jj, self.age = self.ii[~mask], self.jj[~mask], self.age[~mask]

    def rm_keyframe(self, ix):
        TraceGraph.trace.append(("rmkf", int(ix)))
        m = (self.ii == ix)
        self.ii, self.jj, self.age = self.ii[~m], self.jj[~m], self.age[~m]
        self.video.poses[ix] = self.video.poses[ix + 1]
        self.video.disps[ix] = self.video.disps[ix + 1]

    def update(self, t0=None, t1=None, iters=2, use_inactive=False, EPS=1e-7, motion_only=False):
        TraceGraph.trace.append(("update", t0, t1, iters, bool(use_inactive), bool(motion_only)))
        self.age += 1
        n = self.ii.max() + 1
        self.video.poses[1:n] += 0.01                   # any deterministic state change the frontend reads back
        self.video.disps[1:n] *= 1.01


class TraceLoop:
    def __init__(self, net, video, args, cfg):
        pass

    def loop_ba(self, t_start, t_end, steps=6, motion_only=False, local_graph=None):
        TraceGraph.trace.append(("loop_ba", t_start, t_end, steps, bool(motion_only), local_graph is not None))
        return t_end - t_start, 42


FRONTEND_DISTANCES = (0.5, 3.0, 3.0, 0.2, 3.0, 2.3, 2.2, 3.0, 9.0, 0.1, 4.0, 4.0)


def frontend_cfg(enable_loop):
    return {"verbose": False, "tracking": {"warmup": 8, "upsample": True, "beta": 0.75, "frontend": {
        "max_factors": 75, "nms": 1, "keyframe_thresh": 2.25, "window": 10, "thresh": 16.0, "radius": 2,
        "enable_loop": enable_loop}, "backend": backend_cfg()["tracking"]["backend"]}}


def run_frontend_trace(frontend_cls, enable_loop, value_counter):
    """Feeds 8 warm-up keyframes, then one new keyframe per entry of FRONTEND_DISTANCES (the scripted keyframe
    distance decides keep / drop), through a Frontend class; returns (trace, video)."""
    import contextlib
    TraceGraph.trace = []
    B, h, w = 40, 3, 4
    g = torch.Generator().manual_seed(167)
    dq = list(FRONTEND_DISTANCES)

    def distance(ii, jj, beta=0.3, bidirectional=True):
        TraceGraph.trace.append(("distance", [int(x) for x in ii], [int(x) for x in jj], beta, bool(bidirectional)))
        return torch.tensor([dq.pop(0)])
    video = types.SimpleNamespace(
        poses=torch.rand(B, 7, generator=g), disps=torch.rand(B, h, w, generator=g) + 0.5,
        disps_sens=torch.rand(B, h, w, generator=g) * (torch.rand(B, h, w, generator=g) > 0.5),
        timestamp=torch.arange(B).float(), dirty=torch.zeros(B, dtype=torch.bool),
        counter=types.SimpleNamespace(value=0) if value_counter else 0, ready=types.SimpleNamespace(value=0),
        get_lock=contextlib.nullcontext, distance=distance, stereo=False)

    def bump():
        if value_counter:
            video.counter.value += 1
        else:
            video.counter += 1
    fe = frontend_cls(types.SimpleNamespace(update=None), video, types.SimpleNamespace(device="cpu"),
                      frontend_cfg(enable_loop))
    for _ in range(8):
        bump()
        fe()                                             # no-ops until the warm-up count is reached, then initialise
    while dq:
        bump()
        fe()
        TraceGraph.trace.append(("state", fe.t1, fe.count, fe.last_loop_t,
                                 video.counter.value if value_counter else video.counter))
    fe()                                                 # nothing new: must be a no-op
    return TraceGraph.trace, video, fe


def gen_frontend():
    """The reference's Frontend (src/frontend.py:9-160) driven over a scripted keyframe stream with a tracing graph:
    records the exact sequence of graph / loop-closure calls and the video state it leaves."""
    fe = importlib.import_module("refsrc.frontend")
    fe.FactorGraph, fe.LoopClosing = TraceGraph, TraceLoop
    arrays = {}
    for loop in (False, True):
        trace, video, f = run_frontend_trace(fe.Frontend, loop, True)
        tag = "loop" if loop else "noloop"
        arrays[f"{tag}_trace"] = np.array(repr(trace))
        arrays[f"{tag}_poses"], arrays[f"{tag}_disps"], arrays[f"{tag}_dirty"] = video.poses, video.disps, video.dirty
        arrays[f"{tag}_last"] = torch.cat([f.last_pose, f.last_disp.reshape(-1), f.last_time.reshape(-1)])
        arrays[f"{tag}_ready"] = np.array(video.ready.value)
    save("frontend_trace.npz", **arrays)


def droid_modules():
    """wire the reference's sub-modules into the synthetic `refsrc` package and import its droid_net."""
    if "torch_scatter" not in sys.modules:
        ts = types.ModuleType("torch_scatter")
        ts.scatter_mean = None
        sys.modules["torch_scatter"] = ts
    mods = sys.modules["refsrc.modules"]
    mods.GradientClip = importlib.import_module("refsrc.modules.clipping").GradientClip
    mods.ConvGRU = importlib.import_module("refsrc.modules.gru").ConvGRU
    mods.BasicEncoder = importlib.import_module("refsrc.modules.extractor").BasicEncoder
    corr = importlib.import_module("refsrc.modules.corr")
    mods.CorrBlock, mods.AltCorrBlock = corr.CorrBlock, corr.AltCorrBlock
    return importlib.import_module("refsrc.droid_net")


def gen_encoder():
    """The reference's BasicEncoder (src/modules/extractor.py:61-126) in the two configurations DroidNet uses
    (instance norm -> 128 ch, no norm -> 256 ch) on a 2-view 64x96 input, CPU fp32; plus DroidNet's checkpoint keys."""
    dn = droid_modules()
    net = dn.DroidNet().eval()
    sd = named_weights(net.state_dict(), seed=173)
    net.load_state_dict(sd)
    g = torch.Generator().manual_seed(179)
    x = torch.randn(1, 2, 3, 64, 96, generator=g)
    save("encoders.npz", x=x, fnet=net.fnet(x), cnet=net.cnet(x),
         keys=np.array(list(net.state_dict().keys())),
         shapes=np.array([str(tuple(v.shape)) for v in net.state_dict().values()]))


MOTION_SCRIPT = (None, 1.0, 3.1, 0.2, 2.6, 2.4, 9.0)      # mean flow the update operator reports, per frame


def digest(x):
    """small fingerprint of a large tensor: sum, abs-sum and a strided sample"""
    x = torch.as_tensor(x).double().reshape(-1)
    if x.numel() <= 64:
        return x
    return torch.cat([x.sum()[None], x.abs().sum()[None], x[::37][:384]])


def run_motion_filter(mf_cls, net, stereo, value_counter=True):
    """Feeds len(MOTION_SCRIPT) frames through a MotionFilter class whose update operator is scripted; returns the
    list of video.append argument tuples plus the filter's skip counter after each frame."""
    g = torch.Generator().manual_seed(181)
    dq = list(MOTION_SCRIPT[1:])
    appended, counts = [], []

    class Video:
        def __init__(self):
            self.counter = types.SimpleNamespace(value=0) if value_counter else 0

        def append(self, *item):
            appended.append(tuple(x.clone() if torch.is_tensor(x) else x for x in item))
            if value_counter:
                self.counter.value += 1
            else:
                self.counter += 1

    def update(net_, inp_, corr_):
        assert net_.shape == inp_.shape == (1, 1, 128, 16, 16) and corr_.shape == (1, 1, 196, 16, 16)
        m = dq.pop(0)
        delta = torch.zeros(1, 1, 16, 16, 2)
        delta[..., 0] = m
        return net_, delta, torch.ones_like(delta)
    parts = types.SimpleNamespace(cnet=net.cnet, fnet=net.fnet, update=update)
    mf = mf_cls(parts, Video(), thresh=2.5, device="cpu")
    b = 2 if stereo else 1
    for t in range(len(MOTION_SCRIPT)):
        image = torch.rand(b, 3, 128, 128, generator=g)
        depth = torch.rand(128, 128, generator=g) * 4 if t % 4 != 2 else None
        intr = torch.tensor([50.0, 52.0, 48.0, 32.0])
        mf.track(float(t), image, depth, intr, gt_pose=torch.eye(4) * (t + 1))
        counts.append(mf.count)
    return appended, counts


def gen_motion_filter():
    """The reference's MotionFilter.track (src/motion_filter.py:41-90) with its own encoders (CPU fp32) and CorrBlock
    (sampler stood in by the oracle) and a scripted update operator: which frames become keyframes and what is
    appended to the video for each."""
    dn = droid_modules()
    mfm = importlib.import_module("refsrc.motion_filter")
    net = dn.DroidNet().eval()
    net.load_state_dict(named_weights(net.state_dict(), seed=173))
    arrays = {}
    for stereo in (False, True):
        appended, counts = run_motion_filter(mfm.MotionFilter, net, stereo)
        tag = "stereo" if stereo else "mono"
        arrays[f"{tag}_counts"] = np.array(counts)
        arrays[f"{tag}_n"] = np.array(len(appended))
        for k, item in enumerate(appended):
            arrays[f"{tag}_{k}_none"] = np.array([x is None for x in item])
            for a, x in enumerate(item):
                if x is not None:
                    arrays[f"{tag}_{k}_{a}"] = digest(x)
    save("motion_filter.npz", **arrays)


MVF_CASES = (("k3", 3, 12), ("k1", 1, 12), ("kinf", "inf", 12), ("k5_part", 5, 9), ("few", 3, 12))


def make_filter_video(cur_t, seed=191):
    """a 12-keyframe synthetic video laid out like the reference's DepthVideo for MultiviewFilter (full-res buffers
    at the `tiny` 12x16 shape; intrinsics stored at 1/8 scale)."""
    import contextlib
    vid = synth.make_video(12, "tiny", seed=9)
    g = torch.Generator().manual_seed(seed)
    B, (h, w) = 14, vid["disps"].shape[-2:]
    pad = lambda x, fill: torch.cat([x, fill.expand(B - x.shape[0], *x.shape[1:])])
    ident = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])
    pf = pad(vid["poses"], ident[None]).clone()
    pf[:, :3] += 0.05 * torch.randn(B, 3, generator=g)                 # the previously filtered poses differ a bit
    comp = torch.tensor([[0.1, -0.2, 0.05, 0.0, 0.0, 0.0, 1.0]])
    comp[0, 3:] = torch.nn.functional.normalize(torch.tensor([0.02, -0.03, 0.01, 1.0]), dim=0)
    lockable = types.SimpleNamespace(get_lock=contextlib.nullcontext)
    return types.SimpleNamespace(
        counter=types.SimpleNamespace(value=cur_t), get_lock=contextlib.nullcontext, mapping=lockable, scale_factor=8,
        poses=pad(vid["poses"], ident[None]), disps_up=pad(vid["disps"], torch.ones(1, h, w)),
        intrinsics=(vid["intrinsics"][:1] / 8.0).repeat(B, 1), pose_compensate=comp, poses_filtered=pf,
        disps_filtered=torch.zeros(B, h, w), mask_filtered=torch.zeros(B, h, w),
        filtered_id=torch.tensor([-1], dtype=torch.int32), update_priority=torch.rand(B, generator=g),
        bound=torch.zeros(1, 3, 2))


def run_filter_cases(filter_cls):
    out = {}
    for name, kernel, cur_t in MVF_CASES:
        video = make_filter_video(cur_t)
        cfg = {"tracking": {"warmup": 8, "multiview_filter": {
            "thresh": 0.2 if name != "few" else 1e-5, "visible_num": 2 if name != "few" else 6,
            "kernel_size": kernel, "bound_enlarge_scale": 1.1}}}
        slam = types.SimpleNamespace(net=None, video=video, verbose=False, mode="rgbd", H=12, W=16, fx=1, fy=1, cx=1,
                                     cy=1)
        f = filter_cls(cfg, types.SimpleNamespace(device="cpu"), slam)
        f()
        out[name] = {k: getattr(video, k).clone() for k in ("update_priority", "mask_filtered", "disps_filtered",
                                                           "poses_filtered", "filtered_id", "bound")}
        video.counter.value = 8                            # at the warm-up count: must not run
        video.filtered_id[0] = -1
        before = video.bound.clone()
        f()
        assert torch.equal(video.bound, before) and int(video.filtered_id) == -1
    return out


def gen_multiview_filter():
    """The reference's MultiviewFilter.forward (src/multiview_filter.py:99-173) on CPU, with droid_backends.iproj /
    depth_filter stood in by the oracle: masks, filtered disparities / poses, priorities and the scene bound."""
    mvf = importlib.import_module("refsrc.multiview_filter")
    res = run_filter_cases(mvf.MultiviewFilter)
    arrays = {}
    for name, r in res.items():
        for k, v in r.items():
            arrays[f"{name}_{k}"] = v
    save("multiview_filter.npz", **arrays)


FILLER_KF_TIMES = (0.0, 2.0, 5.0, 6.0, 9.0)


def run_filler(filler_cls, net, graph_attr_module, n_frames=19):
    """Runs a PoseTrajectoryFiller class over n_frames frames (one full batch of 16 + a remainder; timestamps before,
    between and after the keyframes) with a recording graph in place of FactorGraph.  Returns (poses [n,7], record)."""
    g = torch.Generator().manual_seed(193)
    rec = {"set": [], "factors": [], "updates": [], "counter": []}

    class Video:
        def __init__(self):
            K = len(FILLER_KF_TIMES)
            self.counter = types.SimpleNamespace(value=K)
            self.timestamp = torch.zeros(40)
            self.timestamp[:K] = torch.tensor(FILLER_KF_TIMES)
            self.poses = torch.zeros(40, 7)
            self.poses[:, 6] = 1.0
            xi = 0.3 * torch.randn(K, 6, generator=g)
            self.poses[:K] = lietorch_shim.SE3.exp(xi).data

        def __setitem__(self, index, item):
            rec["set"].append((index.start, index.stop, [None if x is None else digest(x) for x in item]))
            rec["counter"].append(self.counter.value)
            self.poses[index] = item[2]

    class Graph:
        def __init__(self, video, update_op, **kw):
            self.video = video

        def add_factors(self, ii, jj, remove=False):
            rec["factors"].append((ii.tolist(), jj.tolist()))

        def update(self, t0=None, t1=None, iters=2, use_inactive=False, EPS=1e-7, motion_only=False):
            rec["updates"].append((t0, t1, bool(motion_only), bool(use_inactive)))
            self.video.poses[t0:t1, :3] += 0.01            # stands in for the motion-only BA

    graph_attr_module.FactorGraph = Graph
    video = Video()
    filler = filler_cls(types.SimpleNamespace(cnet=net.cnet, fnet=net.fnet, update=None), video, device="cpu")

    def stream():
        for k in range(n_frames):
            t = -0.5 + 0.55 * k
            depth = torch.rand(64, 96, generator=g) + 1.0
            yield t, torch.rand(1, 3, 64, 96, generator=g), depth, torch.tensor([50.0, 52.0, 48.0, 32.0]), None
    out = filler(stream())
    assert video.counter.value == len(FILLER_KF_TIMES)
    return out.data, rec


def gen_filler():
    """The reference's PoseTrajectoryFiller (src/trajectory_filler.py:8-112) on CPU: keyframe bracketing,
    constant-velocity SE3 interpolation (through the lietorch stand-in), what is parked in the video, the edges and
    the motion-only update calls.  `.cuda()` is made a no-op for the run (the reference hard-codes it)."""
    dn = droid_modules()
    tf = importlib.import_module("refsrc.trajectory_filler")
    net = dn.DroidNet().eval()
    net.load_state_dict(named_weights(net.state_dict(), seed=173))
    keep = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        poses, rec = run_filler(tf.PoseTrajectoryFiller, net, tf)
    finally:
        torch.Tensor.cuda = keep
    arrays = {"poses": poses, "factors": np.array(repr(rec["factors"])), "updates": np.array(repr(rec["updates"])),
              "counter": np.array(rec["counter"]), "n_set": np.array(len(rec["set"]))}
    for k, (a, b, items) in enumerate(rec["set"]):
        arrays[f"set{k}_range"] = np.array([a, b])
        arrays[f"set{k}_none"] = np.array([x is None for x in items])
        for j, x in enumerate(items):
            if x is not None:
                arrays[f"set{k}_{j}"] = x
    save("trajectory_filler.npz", **arrays)


def mapper_cfg():
    return {"mode": "rgbd", "cam": {"H_out": 32, "W_out": 48}, "tracking": {"buffer": 40},
            "mapping": {"device": "cpu", "iters": 2, "decay": 0.5, "w_color_loss": 2.0, "w_sdf_loss": 2.0,
                        "w_eikonal_loss": 0.1, "uncertainty_weight_loss": True, "BA": False, "BA_cam_lr": 1e-3,
                        "pixels": 960, "mapping_window_size": 16, "net_lr": 1e-3, "grid_lr": 1e-2}}


def fill_mapping_video(video, n=30, seed=197):
    """synthetic filtered keyframes in a DepthVideo (the reference's or ours: same buffer names)"""
    g = torch.Generator().manual_seed(seed)
    H, W = 32, 48
    video.images[:n] = torch.rand(n, 3, H, W, generator=g)
    video.disps_filtered[:n] = 0.2 + torch.rand(n, H, W, generator=g)
    video.mask_filtered[:n] = (torch.rand(n, H, W, generator=g) > 0.3).float()
    video.mask_filtered[3] = 0.0                                            # a keyframe with no valid pixel
    video.mask_filtered[3, :2, :30] = 1.0                                   # ... except 60: fewer than 2 x n_rays
    video.poses_filtered[:n] = lietorch_shim.SE3.exp(0.2 * torch.randn(n, 6, generator=g)).data
    video.poses_gt[:n] = torch.eye(4) + 0.01 * torch.randn(n, 4, 4, generator=g)
    video.pose_compensate[0] = lietorch_shim.SE3.exp(0.1 * torch.randn(1, 6, generator=g)).data[0]
    video.update_priority[:n] = torch.rand(n, generator=g)
    video.timestamp[:n] = torch.arange(n).float()
    video.bound[0] = torch.tensor([[-3.0, 3.0], [-2.0, 2.5], [-1.0, 4.0]])


def run_mapper(mapper_cls, video, schedule=((1, False), (5, False), (5, False), (14, False), (30, False), (30, True))):
    """Drives a Mapper class through `schedule` = (filtered_id, the_end) calls with optimize_map replaced by a
    recorder; NumPy and torch RNGs are re-seeded before every call so both implementations draw the same pixels."""
    rec = []

    class Net:
        def __init__(self):
            self.realtime_bound = torch.zeros(3, 2)
            self.p = torch.nn.Parameter(torch.zeros(3))

        def get_training_parameters(self, ignore_keys=()):
            return [self.p]

        def get_volume_parameters(self):
            return []

        def update_bound(self, b):
            self.realtime_bound[:] = b
            rec.append(("bound", digest(b)))

        def to(self, device):
            return self
    import tempfile
    slam = types.SimpleNamespace(verbose=False, bound=torch.zeros(3, 2), video=video, mapping_net=Net(), renderer=None,
                                 reload_map=torch.zeros(1).int(), output=tempfile.mkdtemp(), H=32, W=48, fx=40.0,
                                 fy=41.0, cx=24.0, cy=16.0)
    mapper = mapper_cls(mapper_cfg(), types.SimpleNamespace(device="cpu"), slam)

    def optimize_map(rays_o, rays_d, rays_color, rays_depth, optimizer, num_joint_iters):
        rec.append(("opt", num_joint_iters, tuple(rays_o.shape), digest(rays_o), digest(rays_d), digest(rays_color),
                    digest(rays_depth)))
    mapper.optimize_map = optimize_map
    for k, (fid, the_end) in enumerate(schedule):
        video.filtered_id[0] = fid
        np.random.seed(1000 + k)
        torch.manual_seed(2000 + k)
        mapper(the_end=the_end)
        rec.append(("state", mapper.last_visit, bool(mapper.init), int(slam.reload_map), digest(video.update_priority)))
    return rec, mapper


def flatten_mapper_record(rec):
    """record -> (structure string, list of digests)"""
    struct, nums = [], []
    for r in rec:
        struct.append(tuple(x for x in r if not torch.is_tensor(x)))
        nums += [x for x in r if torch.is_tensor(x)]
    return repr(struct), nums


def gen_mapper():
    """The reference's Mapper.__call__ (src/mapping.py:151-302) on the reference's OWN DepthVideo (CPU): keyframe
    schedule (new / recent / priority / random), priority decay, bound hand-over and every ray batch it assembles."""
    sys.modules["refsrc.geom"].projective_ops = importlib.import_module("refsrc.geom.projective_ops")
    droid_modules()
    dv = importlib.import_module("refsrc.depth_video")
    mp = importlib.import_module("refsrc.mapping")
    torch.autograd.set_detect_anomaly(False)                                # mapping.py switches it on at import
    video = dv.DepthVideo(mapper_cfg(), types.SimpleNamespace(device="cpu"))
    fill_mapping_video(video)
    item = video.get_mapping_item(7, "cpu", decay=1.0)
    rec, mapper = run_mapper(mp.Mapper, video)
    struct, nums = flatten_mapper_record(rec)
    arrays = {"struct": np.array(struct), "n": np.array(len(nums))}
    for k, x in enumerate(nums):
        arrays[f"d{k}"] = x
    for k, x in enumerate(item):
        arrays[f"item7_{k}"] = x
    save("mapper.npz", **arrays)


def gen_mapper_loss():
    """The reference's mapping loss and ITS gradients w.r.t. the renderer's outputs: `Mapper.optimize_map`
    (src/mapping.py:60-137) executed verbatim for one iteration on a fake renderer that hands it leaf tensors, with the
    reference's own `InstantNeuS.compute_sdf_error` (src/InstantNeuS.py:372-400) bound to a bare object.  total_loss is
    caught at its .backward() call.  Rays without depth, samples in front of / around / behind the surface, zero and
    large depth variances."""
    sys.modules["refsrc.geom"].projective_ops = importlib.import_module("refsrc.geom.projective_ops")
    droid_modules()
    dv = importlib.import_module("refsrc.depth_video")
    mp = importlib.import_module("refsrc.mapping")
    neus = importlib.import_module("refsrc.InstantNeuS")
    torch.autograd.set_detect_anomaly(False)
    video = dv.DepthVideo(mapper_cfg(), types.SimpleNamespace(device="cpu"))
    fill_mapping_video(video)
    _, mapper = run_mapper(mp.Mapper, video, schedule=())
    g = torch.Generator().manual_seed(211)
    n, s = 157, 72
    gt = torch.rand(n, generator=g) * 3 + 0.5
    gt[torch.rand(n, generator=g) < 0.2] = 0.0
    z = torch.sort(torch.rand(n, s, generator=g) * 4.5, dim=1).values
    z[:, 30:60] = gt.clamp(min=0.3)[:, None] + (torch.rand(n, 30, generator=g) - 0.5) * 0.3
    col = torch.rand(n, 3, generator=g)
    leaf = lambda t: t.clone().requires_grad_(True)
    dvar = torch.rand(n, 1, generator=g) * 0.1
    dvar[:5] = 0.0                                                          # weight 1 / sqrt(1e-10) = 1e5
    ret = {"color": leaf(torch.rand(n, 3, generator=g)), "depth": leaf(torch.rand(n, 1, generator=g) * 4),
           "depth_variance": leaf(dvar), "sdf": leaf(torch.randn(n, s, generator=g) * 0.2), "z_vals": z,
           "gradient_error": leaf(torch.tensor([0.37]))}
    sdf_obj = types.SimpleNamespace(sdf_truncation=0.16, sdf_sparse_factor=5)

    class Net:
        def to(self, device): return self
        def compute_sdf_error(self, sdf, z_vals, gt_depth):
            return neus.InstantNeuS.compute_sdf_error(sdf_obj, sdf=sdf, z_vals=z_vals, gt_depth=gt_depth)
    mapper.mapping_net = Net()
    mapper.renderer = types.SimpleNamespace(render_batch_ray=lambda **kw: ret)
    mapper.train_params = []
    mapper.verbose = False
    opt = types.SimpleNamespace(zero_grad=lambda: None, step=lambda: None)
    caught = {}
    real_backward = torch.Tensor.backward

    def backward(self, *a, **kw):
        caught["loss"] = self.detach().clone()
        return real_backward(self, *a, **kw)
    torch.Tensor.backward = backward
    try:
        with torch.enable_grad():
            mp.Mapper.optimize_map(mapper, torch.zeros(n, 3), torch.zeros(n, 3), col, gt.clone(), opt, 1)   # (run_mapper put a recorder on the instance)
    finally:
        torch.Tensor.backward = real_backward
    zero = lambda t, like: torch.zeros_like(like) if t is None else t
    save("mapper_loss.npz", rays_color=col, rays_depth=gt, z_vals=z, color=ret["color"].detach(),
         depth=ret["depth"].detach(), depth_variance=ret["depth_variance"].detach(), sdf=ret["sdf"].detach(),
         gradient_error=ret["gradient_error"].detach(), loss=caught["loss"], d_color=ret["color"].grad,
         d_depth=ret["depth"].grad, d_depth_variance=zero(ret["depth_variance"].grad, dvar), d_sdf=ret["sdf"].grad,
         d_gradient_error=ret["gradient_error"].grad,
         weights=torch.tensor([mapper.w_color_loss, mapper.w_sdf_loss, mapper.w_eikonal_loss]))


def gen_reference_kernels():
    """Outputs of the reference's OWN CUDA kernels (oracle/_ref: src/lib/*.cu compiled for the CPU by oracle/build_ref.py) on
    the seeded inputs of tests/test_track_gpu.py -- so that the GPU box can compare the HIP kernels with them without
    loading that module: two `ba` iterations on a 12-keyframe / 40-edge ScanNet-shaped window, frame_distance / projmap /
    iproj / depth_filter on the 10-keyframe tiny video, the fp16 and fp32 lookups."""
    from oracle import build_ref
    build_ref.build()
    R = build_ref.load()
    assert R is not None
    out = {}
    p = synth.make_ba_problem(12, 40, "Scan", 13, True)
    c, _ = DO.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
    p = synth.make_ba_problem(12, 40, "Scan", 13, True, noise_px=0.5, coords=c[0])
    K = p["intrinsics"][0].contiguous()
    po, do = p["poses"].clone(), p["disps"].clone()
    dx, dz = R.ba(po, do, K, p["disps_sens"], p["target"], p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], 2,
                  1e-4, 0.1, False)
    out.update(ba_poses=po, ba_disps=do, ba_dx=dx, ba_dz=dz)
    vid = synth.make_video(10, "tiny", seed=9)
    ii, jj = synth.make_graph(10, 30, seed=9)
    P, D, K = vid["poses"], vid["disps"], vid["intrinsics"][0].contiguous()
    pc, pv = R.projmap(P, D, K, ii, jj)
    out.update(projmap_coords=pc, projmap_valid=pv, iproj=R.iproj(P, D, K),
               frame_distance_03=R.frame_distance(P, D, K, ii, jj, 0.3),
               frame_distance_07=R.frame_distance(P, D, K, ii, jj, 0.7),
               depth_filter=R.depth_filter(P, D, K, torch.tensor([0, 1, 4, 8, 9]),
                                           torch.tensor([0.05, 0.1, 0.2, 0.05, 0.3])))
    # the lookups and one-iteration `ba` (with and without the depth prior) on the inputs of the corresponding GPU tests
    spec = importlib.util.spec_from_file_location("_ttg", os.path.join(os.path.dirname(HERE), "test_track_gpu.py"))
    ttg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ttg)
    coords = ttg._rand_coords(3, 12, 16, 12, 16)
    for tag, dt in (("f16", torch.float16), ("f32", torch.float32)):
        vol = ttg._rand_volume(3, 12, 16, 12, 16, dt)
        out["lookup_" + tag] = R.corr_index_forward(vol, coords, 3)[0].float()
    for tag, rgbd in (("rgbd", True), ("mono", False)):
        q = ttg._ba_problem(DO, 8, 22, "tiny", seed=11, rgbd=rgbd)
        K = q["intrinsics"][0].contiguous()
        po, do = q["poses"].clone(), q["disps"].clone()
        dx, dz = R.ba(po, do, K, q["disps_sens"], q["target"], q["weight"], q["eta"], q["ii"], q["jj"], q["t0"], q["t1"], 1,
                      1e-4, 0.1, False)
        out.update({f"ba1_{tag}_poses": po, f"ba1_{tag}_disps": do, f"ba1_{tag}_dx": dx, f"ba1_{tag}_dz": dz})
    save("reference_kernels.npz", **out)


def gen_reference_kernels_bench():
    """The reference's OWN kernels (oracle/_ref, as gen_reference_kernels) at the BENCH shapes, for the GPU box: two Gauss-Newton
    iterations of `ba_cuda` on the S480 frontend window bench.py times (60x80 maps, P = 25, E = 75, RGB-D; inputs of
    tests/test_benchshape_gpu.py::test_ba_bench_window...) and on the monocular window (Replica 40x80, P = 50, E = 100, no depth
    prior, 6P = 294), `altcorr_forward` (fp32, and fp16 on fp16-representable inputs), `altcorr_backward`'s two gradients and
    `corr_index_backward`.  Disparity maps are kept at every second row / column (the fixture stays below 1 MB); the pose
    rows, dx and a float64 checksum cover the rest."""
    from oracle import build_ref
    build_ref.build()
    R = build_ref.load()
    assert R is not None
    out = {}
    for tag, shape, nkf, ne, rgbd, seed in (("s480", "S480", 25, 75, True, 31), ("mono", "Rep", 50, 100, False, 33)):
        p = synth.make_ba_problem(nkf, ne, shape, seed, rgbd)
        c, _ = DO.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
        p = synth.make_ba_problem(nkf, ne, shape, seed, rgbd, noise_px=0.5, coords=c[0])
        K = p["intrinsics"][0].contiguous()
        po, do = p["poses"].clone(), p["disps"].clone()
        dx, dz = R.ba(po, do, K, p["disps_sens"], p["target"], p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], 2,
                      1e-4, 0.1, False)
        out.update({f"ba_{tag}_poses": po, f"ba_{tag}_dx": dx, f"ba_{tag}_disps_s2": do[:, ::2, ::2].contiguous(),
                    f"ba_{tag}_disps_sum": do.double().sum().reshape(1)})
    # alt-corr (altcorr_kernel.cu:27-149 / :152-290) on the inputs of test_altcorr_forward_matches_oracle
    g = torch.Generator().manual_seed(41)
    B, H1, W1, H2, W2, C, S = 3, 9, 13, 5, 7, 128, 2
    f1 = torch.randn(B, H1, W1, C, generator=g) / 4
    f2 = torch.randn(B, H2, W2, C, generator=g) / 4
    ys, xs = torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32), indexing="ij")
    base = torch.stack([xs * (W2 / W1), ys * (H2 / H1)], -1)
    coords = base[None, None] + 2.0 * torch.randn(B, S, H1, W1, 2, generator=g)
    coords[:, :, 0, 0] = torch.tensor([-9.0, 2.0])
    coords[:, :, 0, 1] = torch.tensor([3.0, 2.0])
    out["altcorr_f32"] = R.altcorr_forward(f1, f2, coords, 3)[0]
    out["altcorr_f16in"] = R.altcorr_forward(f1.half().float(), f2.half().float(), coords, 3)[0]   # fp16-representable inputs
    gc = torch.randn(B, S, 49, H1, W1, generator=g)
    d1, d2, _ = R.altcorr_backward(f1, f2, coords, gc, 3)
    out.update(altcorr_grad=gc, altcorr_d1=d1, altcorr_d2=d2)
    # corr_index_backward (correlation_kernels.cu:74-140)
    spec = importlib.util.spec_from_file_location("_ttg", os.path.join(os.path.dirname(HERE), "test_track_gpu.py"))
    ttg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ttg)
    vol = ttg._rand_volume(2, 5, 6, 9, 11, torch.float32)
    cb = ttg._rand_coords(2, 5, 6, 9, 11, spread=1.5)
    gb = torch.randn(2, 7, 7, 5, 6, generator=torch.Generator().manual_seed(45))
    out.update(lookup_bwd_grad=gb, lookup_bwd=R.corr_index_backward(vol, cb, gb, 3)[0])
    save("reference_kernels_bench.npz", **out)


def gen_reference_altcorr_pyramid():
    """The reference's OWN `AltCorrBlock` (src/modules/corr.py:95-145, imported and run as is: feature pyramid, the per-level
    `pyramid[0][:, ii]` / `pyramid[i][:, jj]` gathers, `coords / 2^i`, the fp32 cast, permutes and the concatenation) over the
    reference's OWN `altcorr_forward_kernel` (src/lib/altcorr_kernel.cu:27-149, compiled for the CPU by oracle/build_ref.py) on
    the 9-edge chunk of synth.make_altcorr_chunk -- what `update_lowmem` evaluates per 13-keyframe chunk, and what `gs_altcorr_pyramid` computes in one
    launch.  Kept: the fp32 result at every third row / column (988 KB) and float64 sums per (edge, level) of the whole map."""
    from oracle import build_ref
    build_ref.build()
    R = build_ref.load()
    assert R is not None
    corr = importlib.import_module("refsrc.modules.corr")
    stub = sys.modules["droid_backends"]
    keep = stub.altcorr_forward
    stub.altcorr_forward = lambda f1, f2, c, r: R.altcorr_forward(f1.contiguous(), f2.contiguous(), c.contiguous(), r)
    try:
        fm, ii, jj, coords = synth.make_altcorr_chunk()
        with torch.no_grad():
            out = corr.AltCorrBlock(fm)(coords, ii, jj)              # [1, 9, 196, 30, 40] fp32
    finally:
        stub.altcorr_forward = keep
    assert out.dtype == torch.float32 and tuple(out.shape) == (1, 9, 196, 30, 40)
    ora = DO.altcorr_lookup(DO.altcorr_pyramid(fm), coords, ii, jj, 3)
    print("reference AltCorrBlock vs oracle restatement: max abs", float((out - ora).abs().max()))
    sums = out[0].double().reshape(9, 4, 49, -1).sum(dim=(2, 3))
    save("reference_altcorr_pyramid.npz", ii=ii, jj=jj, fmaps_sum=fm.double().sum().reshape(1),
         coords_sum=coords.double().sum().reshape(1), corr_s3=out[0, :, :, ::3, ::3].contiguous(), level_sums=sums)


class FakeAltCorr:
    """stands in for AltCorrBlock under update_lowmem (both sides): a deterministic function of its arguments"""
    log = None

    def __init__(self, fmaps, num_levels=4, radius=3):
        FakeAltCorr.log.append(("altcorr_init", tuple(fmaps.shape)))

    def __call__(self, coords, ii, jj):
        FakeAltCorr.log.append(("altcorr", ii.tolist(), jj.tolist(), digest(coords)))
        b, n, h, w, _ = coords.shape
        base = (ii.float() * 0.01 + jj.float() * 0.001).view(1, n, 1, 1, 1)
        return (base + 0.01 * coords.mean(dim=-1, keepdim=True).permute(0, 1, 4, 2, 3)).expand(b, n, 196, h, w).contiguous()


def run_lowmem(graph_cls, scale, value_counter=True):
    """Drives FactorGraph.update_lowmem (the reference's or ours) on a 30-keyframe graph with mocked kernels: a
    deterministic reprojection, FakeAltCorr, a scripted update operator, recording BA / upsample calls.  `scale` = 8 for
    the reference (its graph divides video.ht by 8), 1 for ours.  Returns (log, graph, video)."""
    import contextlib
    log = []
    FakeAltCorr.log = log
    T, h, w = 30, 4, 6
    g = torch.Generator().manual_seed(199)

    def reproject(ii, jj):
        ii, jj = torch.as_tensor(ii), torch.as_tensor(jj)
        yy, xx = torch.meshgrid(torch.arange(h).float(), torch.arange(w).float(), indexing="ij")
        base = torch.stack([xx, yy], -1)[None, None]
        shift = (0.1 * (jj - ii).float() + 0.003 * video.poses[jj, 0] - 0.002 * video.poses[ii, 0]).view(1, -1, 1, 1, 1)
        return base + shift, torch.ones(1, len(ii), h, w, 1)

    def ba(target, weight, damping, ii, jj, t0=1, t1=None, iters=2, lm=1e-4, ep=0.1, motion_only=False, ba_type=None):
        log.append(("ba", digest(target), digest(weight), digest(damping), ii.tolist(), jj.tolist(), t0, t1, iters, lm, ep,
                    bool(motion_only), ba_type, tuple(target.shape), tuple(damping.shape)))
        video.poses[t0:t1, 0] += 0.05 * float(weight.mean())          # the next step's reprojection must see the update

    def upsample(ix, mask):
        log.append(("upsample", ix.tolist(), digest(mask)))

    def update_op(net, inp, corr, flow=None, ii=None, jj=None, **kw):
        log.append(("update_op", ii.tolist(), jj.tolist(), digest(net), digest(inp), digest(corr), digest(flow)))
        n = net.shape[1]
        uniq = torch.unique(ii)
        delta = 0.1 * flow[:, :, :2].permute(0, 1, 3, 4, 2) + 0.01
        weight = torch.sigmoid(corr[:, :, :2].permute(0, 1, 3, 4, 2))
        eta = 0.01 * uniq.float().view(1, -1, 1, 1).expand(1, len(uniq), h, w) + 0.001 * float(net.mean())
        upmask = torch.ones(1, len(uniq), 576, h, w) * uniq.float().view(1, -1, 1, 1, 1)
        return net * 0.9 + 0.01 * inp.float(), delta, weight, eta, upmask
    video = types.SimpleNamespace(
        ht=h * scale, wd=w * scale, stereo=False, counter=types.SimpleNamespace(value=T) if value_counter else T,
        disps=torch.ones(T + 4, h, w), poses=torch.zeros(T + 4, 7), dirty=torch.zeros(T + 4, dtype=torch.bool),
        fmaps=torch.rand(T + 4, 1, 8, h, w, generator=g), nets=torch.rand(T + 4, 8, h, w, generator=g),
        inps=torch.rand(T + 4, 8, h, w, generator=g), reproject=reproject, ba=ba, upsample=upsample,
        get_lock=contextlib.nullcontext)
    video.poses[:, 6] = 1.0
    graph = graph_cls(video, update_op, device="cpu", corr_impl="alt", max_factors=-1, upsample=True)
    # edges: bands |i - j| <= 2 around keyframes 2..11 and 28..29 (three 13-keyframe chunks, the middle one without
    # sources) + three long-range edges
    ii, jj = [], []
    for i in list(range(2, 12)) + list(range(28, 30)):
        for j in range(max(0, i - 2), min(T, i + 3)):
            if i != j:
                ii.append(i); jj.append(j)
    ii += [28, 3, 29]; jj += [3, 28, 5]
    graph.add_factors(torch.tensor(ii), torch.tensor(jj))
    log.append(("edges", graph.ii.tolist(), graph.jj.tolist()))
    graph.update_lowmem(t0=3, t1=T, iters=2, steps=2, max_t=T - 1, ba_type="dense", motion_only=False)
    graph.update_lowmem(steps=1, ba_type="loop", motion_only=True)
    return log, graph, video


def flatten_log(log):
    struct, nums = [], []
    for r in log:
        struct.append(tuple(x for x in r if not torch.is_tensor(x)))
        nums += [x for x in r if torch.is_tensor(x)]
    return repr(struct), nums


def gen_lowmem():
    """The reference's FactorGraph.update_lowmem (src/factor_graph.py:253-321) on CPU with mocked kernels: chunking by
    13 source keyframes, the arguments of every alt-corr / update-operator / upsample / BA call, and the state left."""
    mods = sys.modules["refsrc.modules"]
    corr = importlib.import_module("refsrc.modules.corr")
    mods.CorrBlock, mods.AltCorrBlock = corr.CorrBlock, corr.AltCorrBlock
    fg = importlib.import_module("refsrc.factor_graph")
    keep = fg.AltCorrBlock
    fg.AltCorrBlock = FakeAltCorr
    try:
        log, graph, video = run_lowmem(fg.FactorGraph, 8)
    finally:
        fg.AltCorrBlock = keep
    struct, nums = flatten_log(log)
    arrays = {"struct": np.array(struct), "n": np.array(len(nums)), "net": graph.net, "target": graph.target,
              "weight": graph.weight, "damping": graph.damping, "dirty": video.dirty, "poses": video.poses}
    for k, x in enumerate(nums):
        arrays[f"d{k}"] = x
    save("update_lowmem.npz", **arrays)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is needed to (re)generate the fixtures"
    install_stubs()
    if len(sys.argv) > 1:                                    # python gen_golden.py gen_reference_kernels_bench ...
        for name in sys.argv[1:]:
            globals()[name]()
        sys.exit(0)
    with torch.no_grad():
        gen_update()
        gen_ba()
        gen_graph()
        gen_backend()
        gen_frontend()
        gen_encoder()
        gen_motion_filter()
        gen_multiview_filter()
        gen_filler()
        gen_mapper()
    gen_mapper_loss()
    with torch.no_grad():
        gen_lowmem()
        gen_corr()
        gen_proj()
        gen_render()
        gen_render_mono()
        gen_rays()
    gen_neus()
    gen_neus_backward()
    gen_reference_kernels()
    gen_reference_kernels_bench()
    gen_reference_altcorr_pyramid()
