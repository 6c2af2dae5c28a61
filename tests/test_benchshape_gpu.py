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
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-3, atol=2e-6)
    torch.testing.assert_close(pg.cpu(), po, rtol=0, atol=1e-5)
    if motion_only:
        assert out[1] is None and torch.equal(dg.cpu(), prob["disps"])
    else:
        torch.testing.assert_close(out[1].cpu(), ref[1], rtol=2e-3, atol=1e-5)
        torch.testing.assert_close(dg.cpu(), do, rtol=0, atol=1e-5)


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
