"""Oracle parity at the BASELINE shapes: S480 = 60x80 maps / P=25 / E=75 (the bench workload), Rep =
40x80 (Replica), Scan = 30x40 with a 40-keyframe global-BA graph.  The small-shape tests elsewhere prove the
arithmetic; these prove it where the tolerances are tight: 4800 accumulations per Hessian entry, the single-launch
LDS Cholesky at 6P = 144, fp64-atomic ordering over 75 edges, the wave-cooperative tile8 lookup on full-size
volumes, and the global-BA host plumbing (chunks, index tables, damping rows) end to end."""
import pytest
import torch

from go_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def db(built_lib):
    from go_slam_amd import droid_backends
    return droid_backends


@pytest.fixture(scope="module")
def O():
    from oracle import droid_oracle
    return droid_oracle


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _ba_problem(O, num_kf, num_edges, shape, seed, rgbd=True, noise=0.5):
    p = synth.make_ba_problem(num_kf, num_edges, shape, seed, rgbd)
    c, _ = O.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
    return synth.make_ba_problem(num_kf, num_edges, shape, seed, rgbd, noise_px=noise, coords=c[0])


@pytest.mark.parametrize("shape,rgbd,motion_only", [("S480", True, False), ("S480", False, False), ("Rep", True, False),
                                                    ("S480", True, True)])
def test_ba_at_benchmark_shapes_matches_oracle(db, O, dev, shape, rgbd, motion_only):
    """droid_backends.ba, frontend window P=25 (6P = 144: chol_small_kernel), E=75, 2 GN iterations."""
    prob = _ba_problem(O, 25, 75, shape, seed=101, rgbd=rgbd)
    K = prob["intrinsics"][0].contiguous()
    po, do = prob["poses"].clone(), prob["disps"].clone()
    ref = O.ba(po, do, K, prob["disps_sens"], prob["target"], prob["weight"], prob["eta"], prob["ii"], prob["jj"],
               prob["t0"], prob["t1"], 2, 1e-4, 0.1, motion_only)
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    out = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), prob["t0"], prob["t1"], 2, 1e-4, 0.1,
                motion_only)
    torch.cuda.synchronize()
    assert float(ref[0].abs().max()) > 1e-4, "degenerate problem"
    # dx rtol 1e-4 / atol 1e-6 vs the fp64-accumulated oracle.  Measured at these shapes
    # (tools/ba_error_report.py, profiles/r03_ba_error.json): |dx error| <= 5.7e-7 on |dx| <= 4.5e-2 (relative L2
    # 1.5e-5 ... 5e-5), |dz error| <= 2e-6 on |dz| <= 0.36 (relative L2 <= 3e-5) after two iterations.
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(pg.cpu(), po, rtol=0, atol=2e-6)
    if motion_only:
        assert out[1] is None and torch.equal(dg.cpu(), prob["disps"])
    else:
        torch.testing.assert_close(out[1].cpu(), ref[1], rtol=1e-4, atol=4e-6)
        torch.testing.assert_close(dg.cpu(), do, rtol=0, atol=4e-6)


@pytest.mark.parametrize("shape,n", [("S480", 3), ("Rep", 4)])
def test_production_lookup_tile8_nhwc_bit_exact_vs_oracle(db, O, dev, shape, n):
    """The bench's lookup path -- HIP-built tile8 volume -> corr_pyramid_coop_kernel (NHWC) -- against the oracle's
    lookup of the SAME volume values (the row-major build, separately pinned to the oracle volume within 1 fp16 ulp
    and to the tile8 build bit-exactly), at full map size, windows over every border included."""
    ht, wd, _ = synth.SHAPES[shape]
    f1 = synth.make_features(n, shape, seed=111).to(dev)
    f2 = synth.make_features(n, shape, seed=112).to(dev)
    rm = db.corr_volume_pyramid(f1, f2)
    t8 = db.corr_volume_pyramid(f1, f2, layout=db.CORR_TILE8)
    g = torch.Generator().manual_seed(113)
    ys, xs = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32), indexing="ij")
    base = torch.stack([xs, ys], -1)[None]
    for spread in (2.5, 30.0):
        coords = (base + spread * torch.randn(n, ht, wd, 2, generator=g)).contiguous()
        coords[:, 0, 0] = torch.tensor([-7.25, 3.0])          # wholly outside on the left
        coords[:, 1, 1] = torch.tensor([wd + 2.5, ht - 0.5])   # hanging over the bottom-right corner
        ref = O.corr_lookup([p.cpu() for p in rm], coords[None], 3)[0]
        out = db.corr_lookup_pyramid(t8, coords.to(dev), 3, channels_last=True, layout=db.CORR_TILE8, map_size=(ht, wd))
        assert out.is_contiguous(memory_format=torch.channels_last)
        assert torch.equal(out.cpu(), ref), f"{shape} spread {spread}"


def test_update_operator_fast_path_at_bench_shape_matches_plain_module(built_lib, dev):
    """UpdateModule's production path (own 3x3 convolution, hoisted context gates, HIP epilogues) vs the plain
    nn.Sequential formulation of src/droid_net.py:107-140 under the same autocast, at 60x80 / E = 75 with the bench's
    graph (25 source keyframes)."""
    import bench
    from go_slam_amd.droid_net import UpdateModule
    torch.manual_seed(7)
    cl = torch.channels_last
    op = UpdateModule().to(dev).eval().to(memory_format=cl)
    E, h, w = 75, 60, 80
    mk = lambda c, f: f(torch.randn(E, c, h, w, device=dev)).half().contiguous(memory_format=cl).unsqueeze(0)
    net, inp = mk(128, torch.tanh), mk(128, torch.relu)
    corr = mk(196, lambda t: 0.5 * t)
    motion = torch.randn(1, E, h, w, 4, device=dev).permute(0, 1, 4, 2, 3)
    ii, jj = bench.bench_graph(25, 75, 43)
    ii, jj = ii.to(dev), jj.to(dev)
    outs = []
    for fast in (True, False):
        op.fuse_epilogues = fast
        op.gru.fuse_gates = fast
        op.drop_edge_caches()
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            outs.append([t.float() for t in op(net, inp, corr, motion, ii, jj)])
    for name, a, b in zip(["net", "delta", "weight", "eta", "upmask"], outs[0], outs[1]):
        assert a.shape == b.shape, name
        torch.testing.assert_close(a, b, rtol=5e-3, atol=3e-3, msg=lambda m: f"{name}: {m}")


class _Recorder(torch.nn.Module):
    """Wraps the update operator: records what it was fed and what it returned, per call."""

    def __init__(self, op):
        super().__init__()
        self.op = op
        self.calls = []

    def forward(self, net, inp, corr, motion, ii, jj, **kw):
        out = self.op(net, inp, corr, motion, ii, jj, **kw)
        self.calls.append(dict(corr=corr.float().cpu(), motion=motion.float().cpu(), ii=ii.cpu(), jj=jj.cpu(),
                               delta=out[1].float().cpu(), weight=out[2].float().cpu(), damping=out[3].float().cpu()))
        return out

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.op, name)


def test_global_ba_step_matches_oracle_pipeline(built_lib, O, dev):
    """One FactorGraph.update_lowmem step (src/factor_graph.py:255-321) on a 40-keyframe ScanNet-shaped graph: the
    update operator's outputs are recorded, and everything AROUND it -- reprojection, the 13-keyframe chunking, the
    alt-corr lookups, motion features, target / weight / damping assembly and the dense BA over all edges -- is
    replayed on the CPU oracle from the same initial state.  Poses / disparities after the step must agree."""
    from go_slam_amd.depth_video import DepthVideo
    from go_slam_amd.droid_net import UpdateModule
    from go_slam_amd.factor_graph import FactorGraph
    shape, num_kf, num_edges = "Scan", 40, 140
    ht, wd, _ = synth.SHAPES[shape]
    torch.manual_seed(5)
    vid = synth.make_video(num_kf, shape, seed=5, buffer=num_kf + 4)
    video = DepthVideo(ht, wd, buffer=num_kf + 4, device=dev)
    video.poses.copy_(vid["poses"]); video.disps.copy_(vid["disps"])
    video.disps_sens.copy_(vid["disps_sens"]); video.intrinsics.copy_(vid["intrinsics"])
    video.counter = num_kf
    g = torch.Generator().manual_seed(6)
    fm = torch.randn(num_kf, 128, ht, wd, generator=g).half()
    video.fmaps[:num_kf, 0] = fm.to(dev)
    video.nets[:num_kf] = torch.tanh(torch.randn(num_kf, 128, ht, wd, generator=g)).half().to(dev)
    video.inps[:num_kf] = torch.relu(torch.randn(num_kf, 128, ht, wd, generator=g)).half().to(dev)
    op = UpdateModule().to(dev).eval().to(memory_format=torch.channels_last)
    with torch.no_grad():
        op.delta[2].weight.mul_(0.05); op.delta[2].bias.zero_()
    rec = _Recorder(op)
    graph = FactorGraph(video, rec, device=dev, corr_impl="alt", upsample=False)
    ii, jj = synth.make_graph(num_kf, num_edges, seed=5)
    graph.add_factors(ii.to(dev), jj.to(dev))
    ii_g, jj_g = graph.ii.cpu(), graph.jj.cpu()
    target0 = graph.target.float().cpu().clone()          # [1,E,h,w,2]
    weight0 = graph.weight.float().cpu().clone()
    damping0 = graph.damping.float().cpu().clone()
    poses0, disps0 = video.poses.cpu().clone(), video.disps.cpu().clone()

    graph.update_lowmem(t0=1, t1=num_kf, steps=1, iters=2)
    torch.cuda.synchronize()

    # ---- replay on the oracle
    intr = video.intrinsics.cpu()
    coords1, _ = O.reproject(poses0, disps0, intr, ii_g, jj_g)              # [1,E,h,w,2]
    coords0 = graph.coords0.float().cpu()
    motion = torch.cat([coords1 - coords0, target0 - coords1], -1).permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
    pyr = O.altcorr_pyramid(torch.cat([fm, torch.zeros(2, 128, ht, wd).half()])[None])    # fmaps[:counter + 2]
    target, weight, damping = target0.clone(), weight0.clone(), damping0.clone()
    seen = torch.zeros(ii_g.numel(), dtype=torch.bool)
    calls = list(rec.calls)
    for i in range(int(ii_g.min()), int(ii_g.max()) + 1, 13):           # the reference's chunk rule (:279-283)
        v = (ii_g >= i) & (ii_g < i + 13)
        if int(v.sum()) < 1:
            continue
        call = calls.pop(0)
        sel = v.nonzero().reshape(-1)
        assert torch.equal(call["ii"], ii_g[sel]) and torch.equal(call["jj"], jj_g[sel])
        seen[sel] = True
        ref_corr = O.altcorr_lookup(pyr, coords1[:, sel], ii_g[sel], jj_g[sel], 3)
        torch.testing.assert_close(call["corr"], ref_corr, rtol=2e-3, atol=3e-3)
        torch.testing.assert_close(call["motion"], motion[:, sel], rtol=0, atol=2e-3)
        target[:, sel] = coords1[:, sel] + call["delta"]
        weight[:, sel] = call["weight"]
        damping[torch.unique(ii_g[sel])] = call["damping"].reshape(-1, ht, wd)
    assert not calls, "more operator calls than chunks"
    assert bool(seen.all())
    kx = torch.unique(torch.cat([torch.arange(1, num_kf), ii_g]))
    eta = 0.2 * damping[kx].contiguous() + 1e-7
    tg = target.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
    wg = weight.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
    po, do = poses0.clone(), disps0.clone()
    O.ba(po, do, intr[0].contiguous(), video.disps_sens.cpu(), tg, wg, eta, ii_g, jj_g, 1, num_kf, 2, 1e-5, 1e-2, False)
    do.clamp_(min=0.001)
    assert float((po[1:num_kf] - poses0[1:num_kf]).abs().max()) > 1e-4, "the step must move the poses"
    torch.testing.assert_close(video.poses.cpu(), po, rtol=0, atol=5e-5)
    torch.testing.assert_close(video.disps.cpu(), do, rtol=0, atol=5e-5)


# --------------------------------------------------------------------------------------------------------------------
# Path M (mapping) at the BASELINE batch sizes: 4096 rays x 72 samples per render batch (configs[2]; the reference's
# own mapper batch is 4400 pixels, configs/go_slam.yaml:20) and 32768 rays per mapper step (configs[4]).  This is
# where the small-batch tests of tests/test_neus_gpu.py cannot reach: ~480 fp16 packed-atomic contributions per
# coarse-level table entry under loss scale 128, long runs of equal cells in the wave-level pre-reduction of the
# table gradient, the split-K reductions of the dense gradients, the fused loss kernel on 32768 rays.
# --------------------------------------------------------------------------------------------------------------------
import json
import os


def _record(name, payload):
    """Measured parity numbers go to gpurun_out/ (scratch; the summaries judged are copied to profiles/)."""
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    try:
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "r06_parity.json")
        cur = json.load(open(path)) if os.path.exists(path) else {}
        cur[name] = payload
        json.dump(cur, open(path, "w"), indent=1, sort_keys=True)
    except OSError:
        pass


def _bench_rays(n, seed):
    """the ray distribution of bench.py's NeuS legs: origins inside the bound, 10 % rays without depth"""
    g = torch.Generator().manual_seed(seed)
    o = torch.rand(n, 3, generator=g) * 6 - 3
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[torch.rand(n, generator=g) < 0.1] = 0
    col = torch.rand(n, 3, generator=g)
    pr = torch.rand(24, generator=g)
    return o, d, gt, col, pr


def _load_neus(model, P):
    with torch.no_grad():
        model.sdf_network.encoding.encoding.params.copy_(P["grid"])
        model.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
        model.sdf_network.sdf_layer.bias.copy_(P["sdf_b"])
        model.color_network._B.copy_(P["color_B"])
        model.color_network.network.params.copy_(P["mlp"])
        model.variance_network.variance.fill_(float(P["variance"]))


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


@pytest.fixture(scope="module")
def NO():
    from oracle import neus_oracle
    return neus_oracle


@pytest.fixture(scope="module")
def N(built_lib):
    import go_slam_amd.neus as neus
    return neus


@pytest.fixture
def gather_order(request):
    """run a test with the hashed levels gathered per point (the small-batch order) or LEVEL-MAJOR (the order batches of
    >= 524288 points take: neus_encode_levels_kernel + records) whatever the batch size"""
    from go_slam_amd import _lib
    L = _lib.lib()
    old = L.gs_neus_level_major_min_points(0 if request.param == "level-major" else 1 << 30)
    yield request.param
    L.gs_neus_level_major_min_points(old)


@pytest.mark.parametrize("gather_order", ["per-point", "level-major"], indirect=True)
@pytest.mark.parametrize("grid_init", [0.3, 1e-4])
def test_neus_forward_at_4096_rays_matches_oracle(N, NO, dev, grid_init, gather_order):
    """Renderer.sample + InstantNeuS.forward (src/render.py:99-171, src/InstantNeuS.py:295-370) on ONE 4096-ray x 72
    sample batch -- 294,912 points -- against the CPU oracle: all 9 outputs, in-bound masks exact; in both gather orders."""
    P = NO.make_params(71, grid_init=grid_init)                         # bound [-5, 5]^3 (room0.yaml:4)
    P["rt_bound"] = torch.tensor([[-4.2, 4.6], [-4.4, 4.1], [-3.9, 4.4]])
    o, d, gt, _, pr = _bench_rays(4096, seed=72)
    zr, dr = NO.render_sample(o, d, gt, P["bound"], 24, 48, pr)
    ref = NO.neus_forward(o, d, zr, dr, P)
    model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
    _load_neus(model, P)
    model.update_bound(P["rt_bound"])
    R = N.Renderer(N_samples=24, N_surface=48)
    with torch.no_grad():
        z, dd = R.sample(o.to(dev), d.to(dev), P["bound"].to(dev), gt.to(dev), pr.to(dev))
        assert torch.equal(z.cpu(), zr), "sample placement must be bit-exact"
        out = model(o.to(dev), d.to(dev), z, dd)
    c = {k: v.cpu() for k, v in out.items()}
    assert set(c) == {"color", "depth", "depth_variance", "normal", "weight_sum", "sdf_variance", "sdf", "z_vals",
                      "gradient_error"}
    assert torch.equal(c["z_vals"], ref["z_vals"])
    assert torch.equal(c["sdf"] == 100.0, ref["sdf"] == 100.0), "in-bound masks must agree exactly"
    n_in = int((ref["sdf"] != 100.0).sum())
    assert 0.3 * 4096 * 72 < n_in < 4096 * 72, n_in                     # both branches populated
    torch.testing.assert_close(c["sdf"], ref["sdf"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c["weight_sum"], ref["weight_sum"], rtol=0, atol=5e-4)
    torch.testing.assert_close(c["depth"], ref["depth"], rtol=0, atol=2e-3)
    torch.testing.assert_close(c["depth_variance"], ref["depth_variance"], rtol=1e-2, atol=2e-3)
    torch.testing.assert_close(c["color"], ref["color"], rtol=0, atol=4e-3)
    torch.testing.assert_close(c["normal"], ref["normal"], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(c["gradient_error"], ref["gradient_error"], rtol=2e-3, atol=1e-5)
    torch.testing.assert_close(c["sdf_variance"], ref["sdf_variance"])
    _record(f"forward_4096_grid{grid_init:g}_{gather_order}", {
        "points_in_bound": n_in,
        "max_abs": {k: float((c[k] - ref[k]).abs().max()) for k in ("sdf", "color", "depth", "normal", "weight_sum")}})


@pytest.mark.parametrize("gather_order", ["per-point", "level-major"], indirect=True)
@pytest.mark.parametrize("grad_dtype", [torch.float16, torch.float32])
def test_training_gradients_at_4096_rays_match_autograd_oracle(N, NO, dev, grad_dtype, gather_order):
    """Mapper loss (src/mapping.py:96-132) -> gradient of every trained parameter at the reference's batch size,
    HIP backward vs torch.autograd on the differentiable CPU restatement; both table-gradient modes (tiny-cuda-nn's
    loss-scaled fp16 packed atomics -- the production mode -- and fp32 atomics).  The measured relative L2 error per
    parameter group is recorded (gpurun_out/r03_pathM_parity.json -> profiles/)."""
    from oracle import neus_autograd as NA
    P = NO.make_params(81, grid_init=0.3)
    P["rt_bound"] = torch.tensor([[-4.2, 4.6], [-4.4, 4.1], [-3.9, 4.4]])
    o, d, gt, col, pr = _bench_rays(4096, seed=82)
    z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, pr)
    Pd = {k: (v.clone().requires_grad_(True) if k in ("grid", "sdf_w", "sdf_b", "color_B", "mlp") else v)
          for k, v in P.items()}
    Pd["variance"] = torch.tensor(0.2, requires_grad=True)
    ref_loss = NA.mapping_loss(NA.neus_forward_diff(o, d, z, dist, Pd), col, gt)
    ref_loss.backward()
    model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
    _load_neus(model, P)
    model.update_bound(P["rt_bound"])
    model.grid_grad_dtype = grad_dtype
    out = model(o.to(dev), d.to(dev), z.to(dev), dist.to(dev))
    loss = NA.mapping_loss({k: v for k, v in out.items()}, col.to(dev), gt.to(dev))
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), rtol=2e-3, atol=1e-4)
    pairs = {
        "grid": (model.sdf_network.encoding.encoding.params.grad, Pd["grid"].grad),
        "sdf_w": (model.sdf_network.sdf_layer.weight.grad, Pd["sdf_w"].grad),
        "sdf_b": (model.sdf_network.sdf_layer.bias.grad, Pd["sdf_b"].grad),
        "color_B": (model.color_network._B.grad, Pd["color_B"].grad),
        "mlp": (model.color_network.network.params.grad, Pd["mlp"].grad),
        "variance": (model.variance_network.variance.grad.reshape(1), Pd["variance"].grad.reshape(1)),
    }
    report = {k: _rel(a.cpu().float(), b) for k, (a, b) in pairs.items()}
    # the coarse levels are where the fp16 atomics pile up: level 0 has 17^3 entries for ~200 k in-bound points
    gm = NO.grid_meta()
    gg, gr = pairs["grid"][0].cpu().float().reshape(-1, 2), pairs["grid"][1].reshape(-1, 2)
    per_level = []
    for l in range(16):
        a, b = int(gm["offset"][l]), int(gm["offset"][l]) + int(gm["size"][l])
        per_level.append(_rel(gg[a:b], gr[a:b]))
    _record(f"grad_4096_{str(grad_dtype).split('.')[-1]}", {"rel_l2": report, "grid_rel_l2_per_level": per_level,
                                                           "loss": float(loss.detach()), "ref_loss": float(ref_loss.detach())})
    for k, r in report.items():
        assert r < 5e-3, (report, per_level)
    assert max(per_level) < 2e-2, per_level


def test_fused_mapper_step_at_32768_rays_matches_autograd_path(N, NO, dev):
    """One mapper iteration on configs[4]'s 32768-ray batch (2.36 M points): MapTrainer.step_fused (loss kernel ->
    HIP backward -> flat clip + AdamW) vs the autograd path (reference-style loss, torch clip_grad_norm_ + AdamW) on
    identical models: same loss, same gradients, and the same parameters after the step."""
    from go_slam_amd.neus.distributed import mapping_loss_sharded
    from go_slam_amd.neus.mapper import MapTrainer
    P = NO.make_params(91, grid_init=0.05)
    o, d, gt, col, pr = _bench_rays(32768, seed=92)
    args = [t.to(dev) for t in (o, d, col, gt, pr)]
    pair = []
    for fused in (False, True):
        model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
        _load_neus(model, P)
        if not fused:       # the referee accumulates the table gradient with fp32 atomics (exact to ~1e-5); the fused
            model.grid_grad_dtype = torch.float32       # step uses tiny-cuda-nn's loss-scaled fp16 packed atomics
        pair.append((model, MapTrainer(model, N.Renderer(N_samples=24, N_surface=48), fused=fused)))
    (m_ref, t_ref), (m_fus, t_fus) = pair
    assert t_fus.fused and not t_ref.fused
    R = N.Renderer(N_samples=24, N_surface=48)
    z, dd = R.sample(args[0], args[1], m_ref.bound, args[3], args[4])
    loss_a, _ = mapping_loss_sharded(R.eval_points(args[0], args[1], z, dd, m_ref, None), args[2], args[3],
                                     m_ref.compute_sdf_error)
    loss_a.backward()
    loss_f, grid16, inv_scale = t_fus.fused_gradients(*args)
    torch.testing.assert_close(loss_f.float().cpu(), loss_a.detach().float().cpu(), rtol=2e-4, atol=1e-5)
    ref_g = {"grid": m_ref.sdf_network.encoding.encoding.params.grad, "mlp": m_ref.color_network.network.params.grad,
             "sdf_w": m_ref.sdf_network.sdf_layer.weight.grad, "sdf_b": m_ref.sdf_network.sdf_layer.bias.grad,
             "cB": m_ref.color_network._B.grad, "var": m_ref.variance_network.variance.grad}
    rep = {"grid": _rel(grid16.float().cpu() * inv_scale, ref_g["grid"].cpu())}
    for k in ("mlp", "sdf_w", "sdf_b", "cB", "var"):
        rep[k] = _rel(t_fus.flat.dense_grad(k).cpu(), ref_g[k].reshape(-1).cpu())
    gm = NO.grid_meta()
    gf, gr = (grid16.float().cpu() * inv_scale).reshape(-1, 2), ref_g["grid"].cpu().reshape(-1, 2)
    per_level = [_rel(gf[int(gm["offset"][l]):int(gm["offset"][l]) + int(gm["size"][l])],
                      gr[int(gm["offset"][l]):int(gm["offset"][l]) + int(gm["size"][l])]) for l in range(16)]
    _record("fused_step_32768", {"rel_l2_vs_autograd_path_fp32_atomics": rep, "grid_rel_l2_per_level": per_level,
                                 "loss": float(loss_f)})
    # ~3800 fp16 additions per coarse-level entry: the packed-atomic sum carries fp16 accumulation error (as
    # tiny-cuda-nn's does).  Measured (profiles/r03_pathM_parity.json): 0.75 % relative L2 on level 0 (17^3 entries for
    # 2 M points), 0.2 % on the hashed levels, 0.21 % over the whole table; the dense parameters 1e-7 ... 4e-5.
    assert all(v < 5e-3 for v in rep.values()), (rep, per_level)
    assert max(per_level) < 1.5e-2, per_level
    for p_ in m_ref.parameters():
        p_.grad = None
    l_ref = t_ref.step(*args)
    l_fus = t_fus.step(*args)
    torch.testing.assert_close(l_fus.float().cpu(), l_ref.float().cpu(), rtol=2e-4, atol=1e-5)
    pr_, pf_ = dict(m_ref.named_parameters()), dict(m_fus.named_parameters())
    lr = {"sdf_network.encoding.encoding.params": 1e-2}
    gmax = float(ref_g["grid"].abs().max())
    for k in ["sdf_network.encoding.encoding.params", "sdf_network.sdf_layer.weight", "sdf_network.sdf_layer.bias",
              "color_network._B", "color_network.network.params", "variance_network.variance"]:
        a, b = pf_[k].detach().float().cpu(), pr_[k].detach().float().cpu()
        step = lr.get(k, 1e-3)
        dlt = (a - b).abs()
        assert float(dlt.max()) <= 2 * step * 1.01, (k, float(dlt.max()))
        off = dlt > (2e-5 + 2e-3 * b.abs())
        if k.endswith("encoding.params"):
            # Adam's first step is -lr * sign(g) on every entry: where the gradient is below what fp16 (loss scale 128)
            # resolves, the fp16 sum underflows or lands on the other side of zero and the entry moves differently
            # (measured: 3.2 % of ALL entries, every one of them with |g| < 1e-4 max|g|).  Entries with a gradient that
            # fp16 resolves must move identically.
            big = ref_g["grid"].cpu().abs() > 1e-3 * gmax
            assert int(big.sum()) > 1000
            assert float(off[big].float().mean()) < 1e-3, (float(off[big].float().mean()), float(off.float().mean()))
            assert float(off.float().mean()) < 0.1
        else:
            assert float(off.float().mean()) < 2e-3, (k, float(off.float().mean()))


def test_corr_volume_pyramid_at_S480_matches_oracle(db, O, dev):
    """corr_volume_kernel + pooled levels at the bench map size (4800 x 4800 x 128 per edge) vs CorrBlock.corr /
    avg_pool2d restated on the CPU (src/modules/corr.py:26-41,67-76), row-major AND tile8 layouts."""
    import torch.nn.functional as F
    ht, wd, _ = synth.SHAPES["S480"]
    n = 2
    f1 = synth.make_features(n, "S480", seed=121)
    f2 = synth.make_features(n, "S480", seed=122)
    ref = O.corr_pyramid(f1[None], f2[None])
    for layout in (db.CORR_ROWMAJOR, db.CORR_TILE8):
        out = db.corr_volume_pyramid(f1.to(dev), f2.to(dev), layout=layout)
        if layout == db.CORR_TILE8:
            out = [db.corr_untile8(out[l], ht >> l, wd >> l) if l < 2 else out[l] for l in range(4)]
        o0, r0 = out[0].cpu().float(), ref[0].float()
        assert o0.shape == r0.shape
        diff = (o0 - r0).abs()
        assert float(diff.max()) <= 2 ** -10 * max(1.0, float(r0.abs().max())) * 1.01
        assert float((diff == 0).float().mean()) > 0.97
        for l in range(1, 4):
            low = out[l - 1].cpu()
            hl, wl = low.shape[-2:]
            pooled = F.avg_pool2d(low.reshape(-1, 1, hl, wl).float(), 2, 2).to(torch.float16)
            assert torch.equal(out[l].cpu().reshape(-1, 1, hl // 2, wl // 2), pooled), f"level {l}"
            torch.testing.assert_close(out[l].cpu().float(), ref[l].float().reshape(out[l].shape), rtol=0, atol=2e-3)


def test_update_operator_fast_path_at_bench_shape_matches_fp32_cpu_evaluation(built_lib, dev):
    """The production update operator (own fp16 convolutions with fused ConvGRU gates -- v_rcp / v_exp based sigmoid and
    tanh --, hoisted context gates, fused heads) at 60x80 / E = 75 against the SAME module evaluated in fp32 on the CPU
    (plain nn.Sequential formulation, no library or kernel in common): the referee the GPU-vs-GPU test above lacks.
    Inputs are the fp16-rounded tensors in both runs; tolerance = fp16 rounding of 128 ... 320-term dot products."""
    import bench
    import copy
    from go_slam_amd.droid_net import UpdateModule
    torch.manual_seed(7)
    cl = torch.channels_last
    op = UpdateModule().to(dev).eval().to(memory_format=cl)
    E, h, w = 75, 60, 80
    mk = lambda c, f: f(torch.randn(E, c, h, w, device=dev)).half().contiguous(memory_format=cl).unsqueeze(0)
    net, inp = mk(128, torch.tanh), mk(128, torch.relu)
    corr = mk(196, lambda t: 0.5 * t)
    motion = torch.randn(1, E, h, w, 4, device=dev).clamp(-4, 4).permute(0, 1, 4, 2, 3)
    ii, jj = bench.bench_graph(25, 75, 43)
    with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
        out_g = [t.float().cpu() for t in op(net, inp, corr, motion, ii.to(dev), jj.to(dev))]
    op_c = copy.deepcopy(op).float().cpu()
    op_c.fuse_epilogues = False
    torch.set_num_threads(max(8, torch.get_num_threads()))
    with torch.no_grad():
        out_c = op_c(net.float().cpu().contiguous(), inp.float().cpu().contiguous(), corr.float().cpu().contiguous(),
                     motion.half().float().cpu().contiguous(), ii, jj)
    rep = {}
    for name, a, b in zip(["net", "delta", "weight", "eta", "upmask"], out_g, out_c):
        b = b.float()
        assert a.shape == b.shape, name
        rep[name] = (float((a - b).abs().max()), float((a - b).norm() / b.norm()))
    # fp16 convolutions against fp32: measured on MI355X (profiles/r03_pathM_parity.json "update_operator_vs_fp32")
    # measured: net max |error| 6.2e-4 (relative L2 3.2e-4), delta 4.4e-4, weight 2.1e-4, eta 1.5e-5, upmask 3.5e-4
    assert rep["net"][0] < 3e-3 and rep["net"][1] < 1e-3, rep
    assert rep["delta"][1] < 2e-3 and rep["weight"][1] < 2e-3 and rep["eta"][1] < 2e-3 and rep["upmask"][1] < 2e-3, rep
    _record("update_operator_vs_fp32", {k: {"max_abs": v[0], "rel_l2": v[1]} for k, v in rep.items()})


# ---------------------------------------------------------------------------------------------------------------------
# configs[4] (Replica room0 MONOCULAR: replica_mono.yaml:29-35 window 50 / max_factors 100, :54-55 N_samples 48 /
# N_surface 24) and configs[3] (ScanNet long sequence: >= 200 keyframes in one global BA) at THEIR shapes
# ---------------------------------------------------------------------------------------------------------------------

def _mono_rays(n, seed):
    o, d, gt, col, _ = _bench_rays(n, seed)
    pr = torch.rand(48, generator=torch.Generator().manual_seed(seed + 1))
    return o, d, gt, col, pr


def test_mono_sampling_48_24_bit_exact_and_forward_at_4096_rays(N, NO, dev):
    """the monocular split of the 72 samples -- 48 stratified + 24 near-surface (replica_mono.yaml:54-55;
    src/render.py:99-171) -- through render_sample_wave_kernel: z-values bit-exact vs the oracle, incl. the rays without
    depth (which take N_samples + N_surface stratified samples), then InstantNeuS.forward on the batch."""
    P = NO.make_params(171, grid_init=0.3)
    P["rt_bound"] = torch.tensor([[-4.2, 4.6], [-4.4, 4.1], [-3.9, 4.4]])
    o, d, gt, _, pr = _mono_rays(4096, seed=172)
    model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
    _load_neus(model, P)
    model.update_bound(P["rt_bound"])
    R = N.Renderer(N_samples=48, N_surface=24)
    R0 = N.Renderer(N_samples=48, N_surface=24, perturb=0.0)      # (perturb > 0 draws its own vector when given None)
    for perturb in (None, pr):
        zr, dr = NO.render_sample(o, d, gt, P["bound"], 48, 24, perturb)
        assert zr.shape == (4096, 72)
        with torch.no_grad():
            z, dd = (R0 if perturb is None else R).sample(o.to(dev), d.to(dev), P["bound"].to(dev), gt.to(dev),
                                                           None if perturb is None else perturb.to(dev))
        assert torch.equal(z.cpu(), zr), "48 + 24 sample placement must be bit-exact"
        torch.testing.assert_close(dd.cpu()[:, :-1], dr[:, :-1], rtol=0, atol=0)
        torch.testing.assert_close(dd.cpu()[:, -1], dr[:, -1], rtol=1e-6, atol=0)     # backend-defined mean: 1-2 ulp
    # monocular mapping also renders rays WITHOUT any depth (gt = 0 everywhere falls back to pure stratified sampling)
    z0r, d0r = NO.render_sample(o, d, torch.zeros_like(gt), P["bound"], 48, 24, pr)
    with torch.no_grad():
        z0, _ = R.sample(o.to(dev), d.to(dev), P["bound"].to(dev), torch.zeros_like(gt).to(dev), pr.to(dev))
    assert torch.equal(z0.cpu(), z0r)
    # ... and rays whose box exit lies BEHIND the camera (origin outside the bound, looking away): far clamps to 0 <
    # near, the stratified run is descending and only the reference's sort (:168-171) puts it in order
    o2, d2 = o.clone(), d.clone()
    o2[:64, 0] = 6.0
    d2[:64, 0] = d2[:64, 0].abs() + 0.1
    d2 = torch.nn.functional.normalize(d2, dim=1)
    z2r, d2r = NO.render_sample(o2, d2, gt, P["bound"], 48, 24, pr)
    with torch.no_grad():
        z2, dd2 = R.sample(o2.to(dev), d2.to(dev), P["bound"].to(dev), gt.to(dev), pr.to(dev))
    assert bool((z2r[:64, 1:] >= z2r[:64, :-1]).all()) and torch.equal(z2.cpu(), z2r)
    torch.testing.assert_close(dd2.cpu()[:, :-1], d2r[:, :-1], rtol=0, atol=0)
    ref = NO.neus_forward(o, d, zr, dr, P)
    with torch.no_grad():
        out = model(o.to(dev), d.to(dev), z, dd)
    c = {k: v.cpu() for k, v in out.items()}
    assert torch.equal(c["sdf"] == 100.0, ref["sdf"] == 100.0)
    torch.testing.assert_close(c["sdf"], ref["sdf"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c["depth"], ref["depth"], rtol=0, atol=2e-3)
    torch.testing.assert_close(c["color"], ref["color"], rtol=0, atol=4e-3)
    torch.testing.assert_close(c["normal"], ref["normal"], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(c["weight_sum"], ref["weight_sum"], rtol=0, atol=5e-4)
    _record("mono_forward_4096_48+24", {"max_abs": {k: float((c[k] - ref[k]).abs().max())
                                                     for k in ("sdf", "color", "depth", "normal", "weight_sum")}})


def test_neus_forward_and_loss_at_32768_rays_match_oracle(N, NO, dev):
    """configs[4]'s whole 32768-ray batch (2.36 M points), mono sampling, against the CPU ORACLE (round 3 compared the
    32768-ray step HIP vs HIP only): sdf, in-bound masks, composited outputs, and the mapper loss
    (src/mapping.py:96-132) evaluated on both."""
    from oracle import neus_autograd as NA
    P = NO.make_params(181, grid_init=0.3)
    P["rt_bound"] = torch.tensor([[-4.2, 4.6], [-4.4, 4.1], [-3.9, 4.4]])
    o, d, gt, col, pr = _mono_rays(32768, seed=182)
    zr, dr = NO.render_sample(o, d, gt, P["bound"], 48, 24, pr)
    with torch.no_grad():
        ref = NO.neus_forward(o, d, zr, dr, P)
        ref_loss = NA.mapping_loss(ref, col, gt)
    model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
    _load_neus(model, P)
    model.update_bound(P["rt_bound"])
    R = N.Renderer(N_samples=48, N_surface=24)
    with torch.no_grad():
        z, dd = R.sample(o.to(dev), d.to(dev), P["bound"].to(dev), gt.to(dev), pr.to(dev))
        assert torch.equal(z.cpu(), zr)
        out = model(o.to(dev), d.to(dev), z, dd)
        loss = NA.mapping_loss(dict(out), col.to(dev), gt.to(dev))
    c = {k: v.cpu() for k, v in out.items()}
    assert torch.equal(c["sdf"] == 100.0, ref["sdf"] == 100.0), "in-bound masks must agree exactly"
    torch.testing.assert_close(c["sdf"], ref["sdf"], rtol=1e-4, atol=2e-5)
    torch.testing.assert_close(c["depth"], ref["depth"], rtol=0, atol=2e-3)
    torch.testing.assert_close(c["color"], ref["color"], rtol=0, atol=4e-3)
    torch.testing.assert_close(c["gradient_error"], ref["gradient_error"], rtol=2e-3, atol=1e-5)
    torch.testing.assert_close(loss.cpu(), ref_loss, rtol=1e-3, atol=1e-5)
    _record("mono_forward_32768", {"loss": float(loss), "ref_loss": float(ref_loss),
                                   "max_abs": {k: float((c[k] - ref[k]).abs().max()) for k in ("sdf", "color", "depth")}})


def _ba_vs_oracle(db, O, dev, prob, lm, ep, motion_only=False, iters=2):
    K = prob["intrinsics"][0].contiguous()
    po, do = prob["poses"].clone(), prob["disps"].clone()
    ref = O.ba(po, do, K, prob["disps_sens"], prob["target"], prob["weight"], prob["eta"], prob["ii"], prob["jj"],
               prob["t0"], prob["t1"], iters, lm, ep, motion_only)
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    out = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), prob["t0"], prob["t1"], iters, lm, ep,
                motion_only)
    torch.cuda.synchronize()
    return ref, (po, do), out, (pg.cpu(), dg.cpu())


def _err(a, b):
    return {"max_abs_err": float((a - b).abs().max()), "max_abs": float(b.abs().max()), "rel_l2": _rel(a, b)}


@pytest.mark.parametrize("shape", ["Rep", "S480"])
def test_ba_mono_window_50_keyframes_100_edges_matches_oracle(db, O, dev, shape):
    """The MONOCULAR frontend window (replica_mono.yaml:29,32: window 50, max_factors 100): 6P = 294 unknowns -- past
    the 192 the LDS-resident Cholesky of the RGB-D window holds --, no depth prior (disps_sens = 0), at the Replica map
    size (and the bench's 60x80), SURVEY's rtol 1e-4 / atol 1e-6 on dx."""
    prob = _ba_problem(O, 50, 100, shape, seed=191, rgbd=False)
    assert not bool(prob["disps_sens"].any())
    ref, (po, do), out, (pg, dg) = _ba_vs_oracle(db, O, dev, prob, 1e-4, 0.1)
    assert tuple(out[0].shape) == (49, 6) and float(ref[0].abs().max()) > 1e-4
    _record(f"ba_mono_window_P50_E100_{shape}", {"dx": _err(out[0].cpu(), ref[0]), "dz": _err(out[1].cpu(), ref[1])})
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=2e-6)
    torch.testing.assert_close(out[1].cpu(), ref[1], rtol=1e-4, atol=4e-6)
    torch.testing.assert_close(dg, do, rtol=0, atol=4e-6)


def test_ba_mono_window_cholesky_failure_gives_zero_update(db, O, dev):
    """6P = 294: an indefinite reduced system (negative damping) must leave dx = 0 -- the reference's fallback when
    Eigen's LLT fails (src/lib/droid_kernels.cu:1192-1213) -- on the solver the mono window takes, too."""
    prob = _ba_problem(O, 50, 100, "Scan", seed=192, rgbd=False)
    ref, (po, do), out, (pg, dg) = _ba_vs_oracle(db, O, dev, prob, -2.0, -1.0, iters=1)
    assert not bool(ref[0].any()), "oracle: the factorisation should have failed"
    assert not bool(out[0].cpu().any())
    torch.testing.assert_close(pg, prob["poses"], rtol=0, atol=0)


def test_ba_global_200_keyframes_1200_edges_at_scan_shape_matches_oracle(db, O, dev):
    """BASELINE configs[3]'s stress shape AT its shape (P = 200, E = 1200, 30x40 maps; src/backend.py:96-123,
    src/factor_graph.py:255-321): 6P = 1194 unknowns through the blocked Cholesky, 1200-edge CSR rows, Schur pair
    lists -- dx at SURVEY's rtol 1e-4 / atol 1e-6 (the 'tiny'-map version in test_track_gpu.py runs at 5e-3)."""
    prob = _ba_problem(O, 200, 1200, "Scan", seed=193, rgbd=True)
    ref, (po, do), out, (pg, dg) = _ba_vs_oracle(db, O, dev, prob, 1e-5, 1e-2)
    assert tuple(out[0].shape) == (199, 6) and float(ref[0].abs().max()) > 1e-4
    _record("ba_global_P200_E1200_Scan", {"dx": _err(out[0].cpu(), ref[0]), "dz": _err(out[1].cpu(), ref[1])})
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=2e-6)
    torch.testing.assert_close(out[1].cpu(), ref[1], rtol=1e-4, atol=4e-6)
    torch.testing.assert_close(dg, do, rtol=0, atol=4e-6)


def test_ba_recovers_a_one_metre_trajectory_from_exact_correspondences(db, O, dev):
    """Trajectory parity on a path that actually travels (SURVEY 8d iii): 24 keyframes on a ~1 m arc with 2
    degree rotation steps, correspondences = the EXACT reprojections under the true poses and depths (what a perfect
    update operator would predict), poses started 2 cm / 1 degree off and disparities 5 % off.  Sixteen Gauss-Newton
    iterations of droid_backends.ba (8 calls x iters = 2, as FactorGraph.update drives it) must bring the trajectory back:
    Sim(3)-aligned ATE (src/slam.py:343-360) vs the TRUE trajectory below 1e-4 m, and the HIP trajectory equal to the CPU
    oracle's run of the same loop to 1e-5 m -- on a path where those numbers mean something (the random-weight frontend
    test above moves 7 mm)."""
    import math
    from go_slam_amd.eval_ate import ate_rmse
    N, shape = 24, "Scan"
    vid = synth.make_video(N, shape, seed=301, rgbd=True)
    ht, wd, _ = synth.SHAPES[shape]
    poses_gt, disps_gt, intr = vid["poses"], vid["disps"], vid["intrinsics"]
    pairs = [(i, j) for i in range(N) for j in range(N) if i != j and abs(i - j) <= 3]
    ii = torch.tensor([p[0] for p in pairs])
    jj = torch.tensor([p[1] for p in pairs])
    target, valid = O.reproject(poses_gt, disps_gt, intr, ii, jj)               # exact flow
    target = target[0].permute(0, 3, 1, 2).contiguous()                         # [E,2,h,w]
    weight = valid[0].permute(0, 3, 1, 2).repeat(1, 2, 1, 1).contiguous()       # confident wherever the point is in view
    assert float(weight.mean()) > 0.5
    eta = torch.full((N, ht, wd), 1e-4)
    g = torch.Generator().manual_seed(302)
    p0 = poses_gt.clone()
    p0[1:, :3] += 0.02 * torch.randn(N - 1, 3, generator=g)
    ang = math.radians(1.0) * torch.randn(N - 1, 3, generator=g)
    dq = torch.cat([ang / 2, torch.ones(N - 1, 1)], 1)
    dq = dq / dq.norm(dim=1, keepdim=True)
    q = p0[1:, 3:]
    # quaternion product dq * q ([x, y, z, w])
    x1, y1, z1, w1 = dq.unbind(1)
    x2, y2, z2, w2 = q.unbind(1)
    p0[1:, 3:] = torch.stack([w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2, w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2,
                              w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2, w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2], 1)
    d0 = (disps_gt * (1 + 0.05 * torch.randn(disps_gt.shape, generator=g))).clamp(min=0.05)
    K = intr[0].contiguous()
    sens = disps_gt.clone()

    def centres(p):                                  # camera centres of world->camera poses [t, q]
        from go_slam_amd.lietorch_shim import SE3
        return SE3(p).inv().data[:, :3].double().numpy()

    pg, dg = p0.clone().to(dev), d0.clone().to(dev)
    pc, dc = p0.clone(), d0.clone()
    args_g = [t.to(dev) for t in (K, sens, target, weight, eta, ii, jj)]
    for _ in range(8):
        db.ba(pg, dg, *args_g, 1, N, 2, 1e-4, 0.1, False)
        dg.clamp_(min=0.001)
        O.ba(pc, dc, K, sens, target, weight, eta, ii, jj, 1, N, 2, 1e-4, 0.1, False)
        dc.clamp_(min=0.001)
    torch.cuda.synchronize()
    c_gt, c_g, c_c, c_0 = centres(poses_gt), centres(pg.cpu()), centres(pc), centres(p0)
    path = float(abs(c_gt[1:] - c_gt[:-1]).sum())
    ate0, _ = ate_rmse(c_0, c_gt)
    ate_g, info = ate_rmse(c_g, c_gt)
    ate_c, _ = ate_rmse(c_c, c_gt)
    raw_gc = float(((c_g - c_c) ** 2).sum(1).mean() ** 0.5)
    rec = {"keyframes": N, "edges": len(pairs), "path_length_m": path, "ate_start_m": float(ate0),
           "ate_hip_vs_truth_m": float(ate_g), "ate_oracle_vs_truth_m": float(ate_c), "rmse_hip_vs_oracle_m": raw_gc,
           "max_rel_disparity_err": float(((dg.cpu() - disps_gt).abs() / disps_gt).max())}
    _record("trajectory_recovery_1m", rec)
    assert path > 0.5 and ate0 > 5e-3, rec                        # a real path, a real perturbation
    assert ate_g < 1e-4 and ate_c < 1e-4, rec
    assert raw_gc < 1e-5, rec


# ------------------------------------------------------------------------------------------------------------------
# HIP kernels vs the REFERENCE's own kernels at the bench shapes -- no restatement in between
# ------------------------------------------------------------------------------------------------------------------
def _fixture(name):
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    return {k: torch.from_numpy(np.asarray(g[k])) for k in g.files}


@pytest.mark.parametrize("tag,shape,nkf,ne,rgbd,seed", [("s480", "S480", 25, 75, True, 31), ("mono", "Rep", 50, 100, False, 33)])
def test_ba_at_the_bench_windows_matches_the_reference_kernels(db, O, dev, tag, shape, nkf, ne, rgbd, seed):
    """`droid_backends.ba` (2 Gauss-Newton iterations) against what the reference's `ba_cuda` (src/lib/droid_kernels.cu:1314-1434,
    compiled for the CPU by oracle/build_ref.py, run in the build container by tests/golden/gen_golden.py::
    gen_reference_kernels_bench) produced on the SAME seeded window: the S480 frontend window bench.py times (60x80, P = 25,
    E = 75, RGB-D, 6P = 144: chol_small) and the monocular window (Replica 40x80, P = 50, E = 100, no depth prior, 6P = 294:
    chol_mid).  Tolerance = what the reference's own fp32 block reductions allow: its dx differs from the fp64-accumulated
    oracle's by 1.6e-6 / 3.0e-7 at these windows (profiles/r04_reference_kernels_parity.json), the HIP path from the oracle by
    <= 3.5e-7 (profiles/r04_parity.json) -- dx rtol 1e-4 / atol 3e-6, poses 4e-6, disparities 5e-6 absolute."""
    g = _fixture("reference_kernels_bench.npz")
    prob = _ba_problem(O, nkf, ne, shape, seed=seed, rgbd=rgbd)
    K = prob["intrinsics"][0].contiguous()
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    out = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), prob["t0"], prob["t1"], 2, 1e-4, 0.1, False)
    torch.cuda.synchronize()
    assert db.ba_status(dev)["cholesky_failures"] == 0
    dx, rdx = out[0].cpu(), g[f"ba_{tag}_dx"]
    assert float(rdx.abs().max()) > 5e-3, "degenerate window"
    _record(f"ba_{tag}_window_vs_reference_kernels", {
        "dx": _err(dx, rdx), "poses": _err(pg.cpu(), g[f"ba_{tag}_poses"]),
        "disps_every_second_pixel": _err(dg.cpu()[:, ::2, ::2], g[f"ba_{tag}_disps_s2"]),
        "disps_sum_rel": abs(float(dg.double().sum().cpu()) / float(g[f"ba_{tag}_disps_sum"]) - 1.0)})
    torch.testing.assert_close(dx, rdx, rtol=1e-4, atol=3e-6)
    torch.testing.assert_close(pg.cpu(), g[f"ba_{tag}_poses"], rtol=0, atol=4e-6)
    torch.testing.assert_close(dg.cpu()[:, ::2, ::2], g[f"ba_{tag}_disps_s2"], rtol=0, atol=5e-6)
    assert abs(float(dg.double().sum().cpu()) / float(g[f"ba_{tag}_disps_sum"]) - 1.0) < 1e-6


def test_altcorr_and_lookup_backward_match_the_reference_kernels(db, dev):
    """`altcorr_forward` (fp32; fp16 operands with fp32 accumulation against the reference kernel run on the same
    fp16-representable values), `altcorr_backward` and `corr_index_backward` against the outputs of
    src/lib/altcorr_kernel.cu:27-290 and correlation_kernels.cu:74-140 themselves (reference_kernels_bench.npz)."""
    g = _fixture("reference_kernels_bench.npz")
    gen = torch.Generator().manual_seed(41)
    B, H1, W1, H2, W2, C, S = 3, 9, 13, 5, 7, 128, 2
    f1 = torch.randn(B, H1, W1, C, generator=gen) / 4
    f2 = torch.randn(B, H2, W2, C, generator=gen) / 4
    ys, xs = torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32), indexing="ij")
    base = torch.stack([xs * (W2 / W1), ys * (H2 / H1)], -1)
    coords = base[None, None] + 2.0 * torch.randn(B, S, H1, W1, 2, generator=gen)
    coords[:, :, 0, 0] = torch.tensor([-9.0, 2.0])
    coords[:, :, 0, 1] = torch.tensor([3.0, 2.0])
    out, = db.altcorr_forward(f1.to(dev), f2.to(dev), coords.to(dev), 3)
    torch.testing.assert_close(out.cpu(), g["altcorr_f32"], rtol=1e-5, atol=1e-5)
    out16, = db.altcorr_forward(f1.half().to(dev), f2.half().to(dev), coords.to(dev), 3)
    # fp16 output of fp32-accumulated dot products: half an fp16 ulp of the largest entries (|corr| < 8 -> 2^-8 ulp)
    torch.testing.assert_close(out16.cpu().float(), g["altcorr_f16in"], rtol=2 ** -10, atol=2 ** -12)
    assert torch.equal(torch.randn(B, S, 49, H1, W1, generator=gen), g["altcorr_grad"])
    d1, d2, dc = db.altcorr_backward(f1.to(dev), f2.to(dev), coords.to(dev), g["altcorr_grad"].to(dev), 3)
    torch.testing.assert_close(d1.cpu(), g["altcorr_d1"], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(d2.cpu(), g["altcorr_d2"], rtol=1e-5, atol=2e-5)
    assert not bool(dc.any())
    # corr_index_backward on the volume / coordinates of test_track_gpu.py::test_corr_index_backward_matches_oracle
    import importlib.util
    spec = importlib.util.spec_from_file_location("_ttg", os.path.join(os.path.dirname(__file__), "test_track_gpu.py"))
    ttg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ttg)
    vol = ttg._rand_volume(2, 5, 6, 9, 11, torch.float32)
    cb = ttg._rand_coords(2, 5, 6, 9, 11, spread=1.5)
    vg, = db.corr_index_backward(vol.to(dev), cb.to(dev), g["lookup_bwd_grad"].to(dev), 3)
    torch.testing.assert_close(vg.cpu(), g["lookup_bwd"], rtol=1e-5, atol=1e-6)


def test_altcorr_pyramid_one_launch_matches_the_reference_altcorr_block_on_the_reference_kernel(dev, built_lib):
    """`AltCorrBlock.lookup` = ONE launch of `gs_altcorr_pyramid` (csrc/altcorr_pyramid.hip, what `update_lowmem` runs) against
    the reference's OWN `AltCorrBlock` (src/modules/corr.py:95-145) executed over the reference's OWN `altcorr_forward_kernel`
    (src/lib/altcorr_kernel.cu:27-149, compiled for the CPU) on the same 9-edge ScanNet-shaped chunk
    (tests/golden/gen_golden.py::gen_reference_altcorr_pyramid): smooth flow, 3 px of noise and one edge far outside the map.
    Tolerance: the reference computes in fp32 on the fp16 features, this kernel accumulates fp16 products in fp32 and rounds
    the blended value to fp16 once -- one fp16 ulp of max(1, |corr|) per value; the per-(edge, level) sums over the WHOLE map
    (the fixture keeps every third pixel) to 2e-4 of the level's absolute mass."""
    from go_slam_amd.corr import AltCorrBlock
    g = _fixture("reference_altcorr_pyramid.npz")
    fm, ii, jj, coords = synth.make_altcorr_chunk()
    assert torch.equal(ii, g["ii"]) and torch.equal(jj, g["jj"])
    assert abs(float(fm.double().sum()) - float(g["fmaps_sum"])) < 1e-6 * float(fm.double().abs().sum())
    assert abs(float(coords.double().sum()) - float(g["coords_sum"])) < 1e-6 * float(coords.double().abs().sum())
    blk = AltCorrBlock(fm.to(dev))
    assert blk.fused_supported(coords.to(dev))
    out = blk.lookup(coords.to(dev), ii.to(dev), jj.to(dev))
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1, 9, 196, 30, 40) and out.dtype == torch.float16
    o = out[0].float().cpu()
    ref = g["corr_s3"]
    assert float(ref.abs().max()) > 4.0 and float((ref[:5] != 0).float().mean()) > 0.5, "degenerate fixture"
    ulp = 2.0 ** -10 * ref.abs().clamp(min=1.0)
    err = (o[:, :, ::3, ::3] - ref).abs()
    _record("altcorr_pyramid_vs_reference_altcorr_block", {"max_abs_err": float(err.max()), "max_err_in_ulps": float((err / ulp).max()),
                                                          "ref_max_abs": float(ref.abs().max())})
    assert bool((err <= 1.01 * ulp).all()), float((err / ulp).max())
    sums = o.double().reshape(9, 4, 49, -1).sum(dim=(2, 3))
    mass = o.double().abs().reshape(9, 4, 49, -1).sum(dim=(2, 3)).clamp(min=1.0)
    assert bool(((sums - g["level_sums"]).abs() <= 2e-4 * mass).all()), ((sums - g["level_sums"]).abs() / mass).max()
    # and the per-level entry point (droid_backends.altcorr_forward x 4, the reference's call pattern) on the same chunk
    # (fp16 features give that path fp16 outputs, cast up: one rounding of the same fp32-accumulated value)
    per_level = blk(coords.to(dev), ii.to(dev), jj.to(dev))[0].cpu()
    assert bool(((per_level[:, :, ::3, ::3] - ref).abs() <= 1.01 * ulp).all())


@pytest.mark.parametrize("shape", ["Rep", "S480"])
@pytest.mark.parametrize("case", ["smooth", "noise3px"])
def test_altcorr_pyramid_one_launch_matches_oracle_at_replica_and_bench_maps(O, dev, built_lib, shape, case):
    """The same one-launch kernel at the map sizes `update_lowmem` meets on Replica (40 x 80) and at S480 (60 x 80) -- the
    round-5 tests ran it at 30 x 40 and 10 x 14 only -- against the oracle's restatement of corr.py:112-145 +
    altcorr_kernel.cu:27-149 and against the per-level kernel (itself pinned to the reference kernel's output)."""
    import importlib.util
    import os
    from go_slam_amd.corr import AltCorrBlock
    spec = importlib.util.spec_from_file_location("_ttg", os.path.join(os.path.dirname(__file__), "test_track_gpu.py"))
    ttg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ttg)
    ht, wd, _ = synth.SHAPES[shape]
    g = torch.Generator().manual_seed(223)
    fm = (torch.randn(1, 4, 128, ht, wd, generator=g) * 1.5).half()
    ii = torch.tensor([0, 1, 2, 3, 3, 1, 0, 2, 1])
    jj = torch.tensor([1, 0, 3, 2, 0, 3, 2, 0, 2])
    if case == "smooth":
        coords = ttg._smooth_coords(9, ht, wd, seed=224)
    else:
        coords = ttg._rand_coords(9, ht, wd, ht, wd, seed=225, spread=3.0).permute(0, 2, 3, 1)[None].contiguous()
    ref = O.altcorr_lookup(O.altcorr_pyramid(fm), coords, ii, jj, 3)
    blk = AltCorrBlock(fm.to(dev))
    assert blk.fused_supported(coords.to(dev))
    out = blk.lookup(coords.to(dev), ii.to(dev), jj.to(dev))
    per_level = blk(coords.to(dev), ii.to(dev), jj.to(dev))
    torch.cuda.synchronize()
    assert tuple(out.shape) == (1, 9, 196, ht, wd) and out.dtype == torch.float16
    o, pl = out.float().cpu(), per_level.cpu()
    ulp = 2.0 ** -10 * ref.abs().clamp(min=1.0)
    assert bool(((o - pl).abs() <= 1.01 * ulp).all()), float(((o - pl).abs() / ulp).max())
    assert bool(((o - ref).abs() <= 1.01 * ulp).all()), float(((o - ref).abs() / ulp).max())
    assert bool(((pl - ref).abs() <= 1.01 * ulp).all()), float(((pl - ref).abs() / ulp).max())
