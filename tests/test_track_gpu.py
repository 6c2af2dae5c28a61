"""GPU parity: HIP tracking path (through the C-ABI) vs the CPU oracle on seeded inputs.

Tolerances: lookup fp16 bit-exact (the oracle carries the reference's fp16
accumulation order), fp32 lookups rtol 1e-5; geometry kernels compiled op-by-op
(-ffp-contract=off) => masks / counts bit-exact and coords to ~1 ulp; BA dx rtol 1e-4 /
atol 1e-6, poses/disps after 2 iterations atol 1e-5.
"""
import pytest
import torch

from go_slam_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "gpu tests need an MI355X"
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def db(built_lib):
    from go_slam_amd import droid_backends
    return droid_backends


@pytest.fixture(scope="module")
def O():
    from oracle import droid_oracle
    return droid_oracle


def _rand_volume(n, h1, w1, h2, w2, dtype, seed=0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, h1, w1, h2, w2, generator=g).to(dtype)


def _rand_coords(n, h1, w1, h2, w2, seed=1, spread=6.0):
    """Coordinates around the identity grid with a few far-out / border / integer cases."""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(h1, dtype=torch.float32), torch.arange(w1, dtype=torch.float32),
                            indexing="ij")
    sx, sy = w2 / w1, h2 / h1
    c = torch.stack([xs * sx, ys * sy], 0)[None].repeat(n, 1, 1, 1)
    c = c + spread * torch.randn(n, 2, h1, w1, generator=g)
    c[:, :, 0, 0] = torch.tensor([-20.0, 3.0])[None]          # fully outside
    c[:, :, 0, 1] = torch.tensor([2.0, 2.0])[None]            # exact integer (weights 0/1)
    c[:, :, 1, 0] = torch.tensor([w2 - 1.5, h2 - 0.5])[None]  # straddles the far border
    c[:, :, 1, 1] = torch.tensor([-0.25, -0.75])[None]        # straddles the near border
    return c.contiguous()


@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.float64])
def test_corr_index_forward_matches_oracle(db, O, dev, dtype):
    n, h1, w1, h2, w2 = 3, 12, 16, 12, 16
    vol = _rand_volume(n, h1, w1, h2, w2, dtype)
    coords = _rand_coords(n, h1, w1, h2, w2)
    ref, = O.corr_index_forward(vol, coords, 3)
    out, = db.corr_index_forward(vol.to(dev), coords.to(dev), 3)
    assert out.shape == ref.shape and out.dtype == dtype
    if dtype == torch.float16:
        assert torch.equal(out.cpu(), ref), "fp16 lookup must be bit-identical to the fp16-order oracle"
    else:
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)


def test_corr_index_forward_odd_sizes_and_small_levels(db, O, dev):
    """ScanNet-like floor sizes (30->15->7->3) put most windows across the border."""
    for (h2, w2) in [(7, 10), (3, 5), (15, 20)]:
        vol = _rand_volume(2, 6, 9, h2, w2, torch.float16, seed=h2)
        coords = _rand_coords(2, 6, 9, h2, w2, seed=w2, spread=2.0)
        ref, = O.corr_index_forward(vol, coords, 3)
        out, = db.corr_index_forward(vol.to(dev), coords.to(dev), 3)
        assert torch.equal(out.cpu(), ref)


def test_corr_index_integer_coords_return_the_window(db, dev):
    """KAT: integer coordinates => the 7x7 taps are the volume entries themselves (x-major)."""
    h, w = 12, 16
    vol = torch.arange(h * w, dtype=torch.float32).view(1, 1, 1, h, w).repeat(1, 2, 2, 1, 1).contiguous()
    coords = torch.zeros(1, 2, 2, 2)
    coords[0, 0] = 8.0
    coords[0, 1] = 6.0
    out, = db.corr_index_forward(vol.to(dev), coords.to(dev), 3)
    out = out.cpu()
    for i in range(7):
        for j in range(7):
            assert out[0, i, j, 0, 0] == vol[0, 0, 0, 6 - 3 + j, 8 - 3 + i]


def test_corr_index_generic_radius(db, O, dev):
    vol = _rand_volume(2, 5, 6, 9, 11, torch.float32)
    coords = _rand_coords(2, 5, 6, 9, 11, spread=1.5)
    for r in (1, 2, 4):
        ref, = O.corr_index_forward(vol, coords, r)
        out, = db.corr_index_forward(vol.to(dev), coords.to(dev), r)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)


def test_corr_index_backward_matches_oracle(db, O, dev):
    vol = _rand_volume(2, 5, 6, 9, 11, torch.float32)
    coords = _rand_coords(2, 5, 6, 9, 11, spread=1.5)
    g = torch.randn(2, 7, 7, 5, 6)
    ref, = O.corr_index_backward(vol, coords, g, 3)
    out, = db.corr_index_backward(vol.to(dev), coords.to(dev), g.to(dev), 3)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("shape", ["tiny", "Scan"])
def test_corr_lookup_pyramid_matches_four_level_loop(db, O, dev, shape):
    ht, wd, _ = synth.SHAPES[shape]
    n = 3
    f1 = synth.make_features(n, shape, seed=5)[None]
    f2 = synth.make_features(n, shape, seed=6)[None]
    pyr = O.corr_pyramid(f1, f2)
    coords = _rand_coords(n, ht, wd, ht, wd, spread=4.0).permute(0, 2, 3, 1).contiguous()
    ref = O.corr_lookup(pyr, coords[None], 3)[0]
    out = db.corr_lookup_pyramid([p.to(dev) for p in pyr], coords.to(dev), 3)
    assert out.dtype == torch.float16 and tuple(out.shape) == (n, 196, ht, wd)
    assert torch.equal(out.cpu(), ref)
    # same logical tensor, NHWC strides (what the update operator consumes)
    out_cl = db.corr_lookup_pyramid([p.to(dev) for p in pyr], coords.to(dev), 3, channels_last=True)
    assert out_cl.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(out_cl.cpu(), ref)


def test_reproject_matches_oracle_and_identity_kat(db, O, dev):
    vid = synth.make_video(8, "tiny", seed=7)
    ii, jj = synth.make_graph(8, 24, seed=7)
    ii = torch.cat([ii, torch.tensor([2, 5])])       # two stereo (i == j) edges
    jj = torch.cat([jj, torch.tensor([2, 5])])
    ref_c, ref_v = O.reproject(vid["poses"], vid["disps"], vid["intrinsics"], ii, jj)
    c, v = db.reproject(vid["poses"].to(dev), vid["disps"].to(dev), vid["intrinsics"].to(dev), ii.to(dev), jj.to(dev))
    assert torch.equal(v.cpu(), ref_v)
    torch.testing.assert_close(c.cpu(), ref_c, rtol=1e-6, atol=1e-4)
    # identical poses => coordinates are the pixel grid
    vid["poses"][:] = vid["poses"][0]
    c, v = db.reproject(vid["poses"].to(dev), vid["disps"].to(dev), vid["intrinsics"].to(dev),
                        ii[:24].to(dev), jj[:24].to(dev))
    ys, xs = torch.meshgrid(torch.arange(12.0), torch.arange(16.0), indexing="ij")
    torch.testing.assert_close(c.cpu()[0], torch.stack([xs, ys], -1)[None].expand(24, -1, -1, -1), atol=2e-4, rtol=0)


def _fixture(name):
    import os
    import numpy as np
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", name))
    return {k: torch.from_numpy(np.asarray(g[k])) for k in g.files if g[k].dtype.kind in "fiub"}


def test_reproject_matches_the_reference_fixture(db, dev):
    """gs_reproject against tests/golden/proj.npz = the REFERENCE's `projective_transform` (src/geom/projective_ops.py:114-144)
    executed verbatim (incl. a stereo edge) -- the fixture the oracle is pinned to, here without the oracle in between."""
    g = _fixture("proj.npz")
    c, v = db.reproject(g["poses"].to(dev), g["disps"].to(dev), g["intrinsics"].to(dev), g["ii"].to(dev), g["jj"].to(dev))
    assert torch.equal(v.cpu().reshape(g["valid"].shape), g["valid"])
    torch.testing.assert_close(c.cpu().reshape(g["coords"].shape), g["coords"], rtol=1e-6, atol=1e-4)


def test_ba_pose_update_matches_the_reference_python_ba_fixture(db, dev):
    """One Gauss-Newton step of `droid_backends.ba` against tests/golden/ba_python.npz = the reference's OWN pure-PyTorch
    bundle adjustment (src/geom/ba.py + src/geom/chol.py) run on the CPU: the POSE update (accumulation, Schur
    complement, damped Cholesky solve, retraction) must agree to the fp32 accuracy of the Python's normal equations.  The
    depth update is not compared here: the CUDA kernel this path mirrors skips `pose index <= 0` terms in the depth
    back-substitution, the Python does not (tests/test_oracle_pinned.py compares it with that quirk switched off in the
    oracle; the quirk itself is checked against the oracle in the BA parity tests)."""
    G = _fixture("ba_python.npz")
    poses, disps = G["poses"].clone().to(dev), G["disps"].clone().to(dev)
    n = poses.shape[0]
    K = G["intrinsics"][0].contiguous().to(dev)
    target = G["target"].permute(0, 3, 1, 2).contiguous().to(dev)
    weight = G["weight"].permute(0, 3, 1, 2).contiguous().to(dev)
    db.ba(poses, disps, K, torch.zeros_like(disps), target, weight, (G["eta"] + 1e-7).to(dev), G["ii"].to(dev),
          G["jj"].to(dev), 1, n, 1, 1e-4, 0.1, False)
    assert float((G["poses_out"] - G["poses"]).abs().max()) > 1e-4          # the step is not a no-op
    torch.testing.assert_close(poses.cpu(), G["poses_out"], rtol=0, atol=1e-4)
    assert db.ba_status(dev)["cholesky_failures"] == 0


@pytest.mark.parametrize("tag,dt", [("f32", torch.float32), ("f16", torch.float16)])
def test_corrblock_matches_the_reference_fixture(dev, built_lib, tag, dt):
    """`CorrBlock(fmap1, fmap2)(coords)` against tests/golden/corr_*.npz = the reference's CorrBlock (src/modules/corr.py:26-53,
    67-76) executed verbatim: the four pyramid levels and the 196-channel lookup.  fp16 takes the MFMA volume kernel (the
    GEMM's fp32 accumulation order may differ by one fp16 rounding per entry, as for the oracle), fp32 the reference's own
    matmul + avg_pool2d formulation followed by the HIP lookup."""
    from go_slam_amd.corr import CorrBlock
    g = _fixture(f"corr_{tag}.npz")
    blk = CorrBlock(g["fmap1"].to(dt).to(dev), g["fmap2"].to(dt).to(dev))
    for i in range(4):
        ref = g[f"pyr{i}"]
        got = blk.corr_pyramid[i].float().cpu().reshape(ref.shape)
        if dt == torch.float32:
            torch.testing.assert_close(got, ref, rtol=1e-5, atol=1e-5)
        else:
            diff = (got - ref).abs()
            assert float(diff.max()) <= 2 ** -10 * max(1.0, float(ref.abs().max())), (i, float(diff.max()))
            assert float((diff == 0).float().mean()) > 0.98, (i, float((diff == 0).float().mean()))
    out = blk(g["coords"].to(dev)).float().cpu()
    ref = g["lookup"]
    assert out.shape == ref.shape
    if dt == torch.float32:
        torch.testing.assert_close(out, ref, rtol=1e-5, atol=1e-5)
    else:       # built from a volume that may differ by one rounding in < 2 % of its entries
        diff = (out - ref).abs()
        assert float(diff.max()) <= 2 ** -9 * max(1.0, float(ref.abs().max())), float(diff.max())
        assert float((diff == 0).float().mean()) > 0.95, float((diff == 0).float().mean())


def test_projmap_frame_distance_iproj_depth_filter(db, O, dev):
    vid = synth.make_video(10, "tiny", seed=9)
    ii, jj = synth.make_graph(10, 30, seed=9)
    P, D, K = vid["poses"], vid["disps"], vid["intrinsics"][0].contiguous()
    Pd, Dd, Kd = P.to(dev), D.to(dev), K.to(dev)
    # projmap
    rc, rv = O.projmap(P, D, K, ii, jj)
    c, v = db.projmap(Pd, Dd, Kd, ii.to(dev), jj.to(dev))
    assert torch.equal(v.cpu(), rv)
    torch.testing.assert_close(c.cpu(), rc, rtol=1e-6, atol=1e-4)
    # frame_distance (reduction order differs: rtol 1e-4)
    for beta in (0.3, 0.7):
        ref = O.frame_distance(P, D, K, ii, jj, beta)
        out = db.frame_distance(Pd, Dd, Kd, ii.to(dev), jj.to(dev), beta)
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    # a pair that looks away => < 75 % valid => 1000
    P2 = P.clone()
    P2[1, 3:] = torch.tensor([0.0, 1.0, 0.0, 0.0])       # 180 deg about y
    out = db.frame_distance(P2.to(dev), Dd, Kd, torch.tensor([0]).to(dev), torch.tensor([1]).to(dev), 0.3)
    ref = O.frame_distance(P2, D, K, torch.tensor([0]), torch.tensor([1]), 0.3)
    assert float(out) == float(ref)
    # iproj
    ref = O.iproj(P, D, K)
    out = db.iproj(Pd, Dd, Kd)
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-6, atol=1e-5)
    # depth_filter: integer counts, bit-exact; includes frames whose neighbours fall off both ends
    ix = torch.tensor([0, 1, 4, 8, 9])
    th = torch.tensor([0.05, 0.1, 0.2, 0.05, 0.3])
    ref = O.depth_filter(P, D, K, ix, th)
    out = db.depth_filter(Pd, Dd, Kd, ix.to(dev), th.to(dev))
    assert torch.equal(out.cpu(), ref)
    assert ref.max() >= 1


def _run_ba_pair(db, O, dev, prob, iters, lm, ep, motion_only):
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


def _ba_problem(O, num_kf, num_edges, shape, seed, rgbd=True, noise=0.5):
    p = synth.make_ba_problem(num_kf, num_edges, shape, seed, rgbd)
    c, _ = O.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
    return synth.make_ba_problem(num_kf, num_edges, shape, seed, rgbd, noise_px=noise, coords=c[0])


@pytest.mark.parametrize("rgbd", [True, False])
def test_ba_one_iteration_matches_oracle(db, O, dev, rgbd):
    prob = _ba_problem(O, 8, 22, "tiny", seed=11, rgbd=rgbd)
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 1, 1e-4, 0.1, False)
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(out[1].cpu(), ref[1], rtol=1e-3, atol=1e-5)
    torch.testing.assert_close(pg, po, rtol=0, atol=1e-5)
    torch.testing.assert_close(dg, do, rtol=0, atol=1e-5)


@pytest.mark.parametrize("rgbd,motion_only", [(True, False), (False, False), (True, True)])
def test_ba_with_long_out_edge_lists_takes_the_split_accumulation(db, O, dev, rgbd, motion_only):
    """The live frontend's BA runs over the window's active AND inactive edges: ~8 out-edges per keyframe, lists of 20.  From
    6 edges per depth keyframe on, `gs_ba` deals every keyframe's list to 4 workgroups (`ba_accum_kernel`, gridDim.z = 4) and
    adds the per-pixel partial sums in `ba_accum_finish_kernel`; below, one workgroup per keyframe finishes by itself.  Here:
    12 keyframes / 110 edges at the ScanNet map size (9.2 edges per keyframe, lists from 2 to 19 long -- unsplit lists,
    lists shorter than the split count and lists of several rounds in ONE launch) against the fp64-accumulated oracle at
    SURVEY's tolerance, two Gauss-Newton iterations; motion-only never splits (no per-pixel sums exist)."""
    prob = _ba_problem(O, 12, 110, "Scan", seed=17, rgbd=rgbd)
    keep = torch.ones(110, dtype=torch.bool)                 # thin two keyframes' lists to 2 and 1 edges
    for kf, left in ((3, 2), (5, 1)):
        idx = torch.nonzero(prob["ii"] == kf).reshape(-1)
        keep[idx[left:]] = False
    for k in ("ii", "jj", "target", "weight"):
        prob[k] = prob[k][keep].contiguous()
    deg = torch.bincount(prob["ii"], minlength=12)
    assert prob["ii"].numel() >= 6 * 12 and int(deg.max()) >= 9 and sorted(deg.tolist())[:2] == [1, 2], deg.tolist()
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-4, 0.1, motion_only)
    assert float(ref[0].abs().max()) > 1e-4
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=2e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=1e-5)
    if motion_only:
        assert out[1] is None and torch.equal(dg, prob["disps"])
    else:
        torch.testing.assert_close(out[1].cpu(), ref[1], rtol=1e-3, atol=1e-5)
        torch.testing.assert_close(dg, do, rtol=0, atol=1e-5)
    assert db.ba_status(dev)["cholesky_failures"] == 0


def test_ba_two_iterations_frontend_like(db, O, dev):
    prob = _ba_problem(O, 12, 40, "Scan", seed=13)
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-4, 0.1, False)
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-3, atol=2e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=1e-5)
    torch.testing.assert_close(dg, do, rtol=0, atol=1e-5)


def test_ba_motion_only(db, O, dev):
    prob = _ba_problem(O, 8, 22, "tiny", seed=17)
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-4, 0.1, True)
    assert out[1] is None
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=1e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=1e-5)
    assert torch.equal(dg, prob["disps"]), "motion_only must not touch disparities"


def test_ba_window_and_stereo_edges(db, O, dev):
    """t0 > 1 (fixed early frames stay in the graph) plus i==j stereo edges (Q4) and the
    EvT `<= 0` skip (Q1) are all exercised by this graph."""
    prob = _ba_problem(O, 10, 30, "tiny", seed=19)
    prob["ii"] = torch.cat([prob["ii"], torch.tensor([3, 6])])
    prob["jj"] = torch.cat([prob["jj"], torch.tensor([3, 6])])
    g = torch.Generator().manual_seed(3)
    ht, wd, _ = synth.SHAPES["tiny"]
    prob["target"] = torch.cat([prob["target"], prob["target"][:2] + 0.1], 0).contiguous()
    prob["weight"] = torch.cat([prob["weight"], torch.rand(2, 2, ht, wd, generator=g)], 0).contiguous()
    prob["t0"], prob["t1"] = 3, 10
    kx = torch.unique(torch.cat([torch.arange(3, 10), prob["ii"]]))
    prob["eta"] = (1e-2 * torch.rand(len(kx), ht, wd, generator=g) + 1e-4)
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-4, 0.1, False)
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-3, atol=2e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=1e-5)
    torch.testing.assert_close(dg, do, rtol=0, atol=1e-5)
    assert torch.equal(pg[:3], prob["poses"][:3]), "poses before t0 are fixed"


def test_ba_zero_residual_is_a_fixed_point(db, O, dev):
    prob = _ba_problem(O, 6, 14, "tiny", seed=23, rgbd=False, noise=0.0)
    K = prob["intrinsics"][0].contiguous()
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    dx, dz = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                   prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), 1, 6, 2, 1e-4, 0.1, False)
    assert dx.abs().max() < 1e-4 and dz.abs().max() < 1e-3


def test_ba_cholesky_failure_gives_zero_update(db, dev):
    """Negative damping makes the system indefinite => LLT fails => dx = 0 (reference
    droid_kernels.cu:1202-1210), poses untouched."""
    from oracle import droid_oracle as O
    prob = _ba_problem(O, 6, 14, "tiny", seed=29)
    K = prob["intrinsics"][0].contiguous()
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    dx, _ = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                  prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), 1, 6, 1, -2.0, -1.0, True)
    assert torch.count_nonzero(dx) == 0
    assert torch.equal(pg.cpu(), prob["poses"])


def test_ba_status_word_reports_failures_and_depth_row_mismatch(db, dev):
    """droid_backends.ba_status(): Cholesky failures are counted; an `eta` whose row count differs from the number of
    depth keyframes of the graph is flagged -- surplus rows of dz come back ZERO (not uninitialised), missing rows are
    dropped -- and GOSLAM_BA_CHECK turns the flag into the shape error the reference would raise."""
    from oracle import droid_oracle as O
    prob = _ba_problem(O, 6, 14, "tiny", seed=29)
    K = prob["intrinsics"][0].contiguous().to(dev)
    pad = lambda x: torch.cat([x, x[-2:]], 0).contiguous()              # two frames no edge refers to (nbuf = 8)

    def args(eta):
        return (pad(prob["poses"]).to(dev), pad(prob["disps"]).to(dev), K, pad(prob["disps_sens"]).to(dev),
                prob["target"].to(dev), prob["weight"].to(dev), eta.to(dev), prob["ii"].to(dev), prob["jj"].to(dev), 1, 6)

    M = prob["eta"].shape[0]
    db.ba(*args(prob["eta"]), 2, 1e-4, 0.1, False)
    assert db.ba_status(dev) == {"depth_keyframes": M, "depth_rows_mismatch": False, "cholesky_failures": 0}
    db.ba(*args(prob["eta"]), 1, -2.0, -1.0, True)                      # indefinite system
    assert db.ba_status(dev)["cholesky_failures"] == 1
    eta_big = torch.cat([prob["eta"], prob["eta"][:2]], 0).contiguous()
    dx, dz = db.ba(*args(eta_big), 1, 1e-4, 0.1, False)
    st = db.ba_status(dev)
    assert st["depth_rows_mismatch"] and st["depth_keyframes"] == M
    assert dz.shape[0] == M + 2 and torch.count_nonzero(dz[M:]) == 0 and torch.isfinite(dz).all()
    assert torch.count_nonzero(dz[:M]) > 0
    db.ba(*args(prob["eta"][:-1].contiguous()), 1, 1e-4, 0.1, False)    # too few rows
    assert db.ba_status(dev)["depth_rows_mismatch"]
    keep = db.BA_CHECK
    try:
        db.BA_CHECK = True
        with pytest.raises(RuntimeError, match="depth keyframes"):
            db.ba(*args(eta_big), 1, 1e-4, 0.1, False)
    finally:
        db.BA_CHECK = keep


def test_ba_reuses_index_tables_on_one_edge_set(db, dev):
    """droid_backends.ba(..., tables=dict): the second and third call on the same edge tensors skip the table kernel
    (GS_BA_REUSE_TABLES) and must produce exactly what three plain calls produce -- poses / disparities move between the
    calls, the tables depend on the edge set and window only; a different window rebuilds them."""
    from oracle import droid_oracle as O
    prob = _ba_problem(O, 8, 22, "tiny", seed=37)
    K = prob["intrinsics"][0].contiguous().to(dev)
    fixed = [prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev), prob["eta"].to(dev),
             prob["ii"].to(dev), prob["jj"].to(dev)]

    def run(tables_of):
        pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
        outs = []
        for it in range(3):
            outs.append(db.ba(pg, dg, K, *fixed, prob["t0"], prob["t1"], 2, 1e-4, 0.1, False, **tables_of(it)))
        return pg, dg, outs

    pa, da, oa = run(lambda it: {})
    cache = {}
    pb, dbb, ob = run(lambda it: {"tables": cache})
    assert "workspace" in cache and cache["key"][4] == prob["ii"].numel()
    assert torch.equal(pa, pb) and torch.equal(da, dbb)
    for (dxa, dza), (dxb, dzb) in zip(oa, ob):
        assert torch.equal(dxa, dxb) and torch.equal(dza, dzb)
    assert db.ba_status(dev)["cholesky_failures"] == 0
    # another window on the same cache object: tables are rebuilt (key mismatch), result = plain call
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    ws_before = cache["workspace"].data_ptr()
    kx = torch.unique(torch.cat([torch.arange(2, prob["t1"]), prob["ii"]]))
    eta2 = prob["eta"][: len(kx)].contiguous().to(dev) if len(kx) <= prob["eta"].shape[0] else fixed[3]
    f2 = [fixed[0], fixed[1], fixed[2], eta2, fixed[4], fixed[5]]
    r1 = db.ba(pg, dg, K, *f2, 2, prob["t1"], 1, 1e-4, 0.1, False, tables=cache)
    pg2, dg2 = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    r2 = db.ba(pg2, dg2, K, *f2, 2, prob["t1"], 1, 1e-4, 0.1, False)
    assert cache["key"][5] == 2 and torch.equal(r1[0], r2[0]) and torch.equal(pg, pg2)
    del ws_before


def test_ba_rejects_non_contiguous(db, dev):
    from oracle import droid_oracle as O
    prob = _ba_problem(O, 6, 14, "tiny", seed=31)
    K = prob["intrinsics"][0].contiguous().to(dev)
    tgt = prob["target"].to(dev).permute(0, 1, 3, 2)
    with pytest.raises(RuntimeError, match="targets must be contiguous"):
        db.ba(prob["poses"].to(dev), prob["disps"].to(dev), K, prob["disps_sens"].to(dev), tgt,
              prob["weight"].to(dev), prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), 1, 6, 1, 1e-4, 0.1,
              False)


# 6P -> solver (csrc/chol.hip, gs_chol_solve_launch): <= 192 chol_small (LDS-resident); 193 .. 300 chol_mid, ONE workgroup with
# head stages of 60 columns (mid_tile_product<15>); above 300 the multi-kernel blocked path (one launch per 32-column panel:
# chol_panel_kernel, then chol_step_kernel = the next panel beside the previous panel's trailing update).  Every branch and
# both of its edges, incl. stage counts that leave a ragged last stage (n % 60 != 0) and panel counts that leave a ragged
# last panel (6P % 32 != 0):
@pytest.mark.parametrize("num_kf,path", [(34, "mid, one head stage (6P = 198: six columns past chol_small's cap)"),
                                         (48, "mid (6P = 282, 282 % 60 = 42)"),
                                         (51, "mid at its upper edge (6P = 300)"),
                                         (52, "blocked multi-kernel path at its lower edge (6P = 306, 306 % 32 = 18)"),
                                         (58, "blocked multi-kernel path (6P = 342, 342 % 32 = 22)"),
                                         (61, "blocked multi-kernel path (6P = 360, 360 % 32 = 8)"),
                                         (65, "blocked multi-kernel path, whole panels (6P = 384 = 12 x 32)"),
                                         (76, "blocked multi-kernel path (6P = 450, last panel of 2 columns)"),
                                         (77, "blocked multi-kernel path (6P = 456)")])
def test_ba_cholesky_path_for_every_window_size(db, O, dev, num_kf, path):
    """Two Gauss-Newton iterations of `ba` vs the oracle with the pose system going through each solver branch."""
    prob = _ba_problem(O, num_kf, int(5.4 * num_kf), "tiny", seed=37 + num_kf)
    assert 6 * (prob["t1"] - prob["t0"]) == 6 * (num_kf - 1), path
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-5, 1e-2, False)
    assert db.ba_status(dev)["cholesky_failures"] == 0, path
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=2e-3, atol=5e-6, msg=lambda m: path + ": " + m)
    torch.testing.assert_close(pg, po, rtol=0, atol=2e-5, msg=lambda m: path + ": " + m)
    torch.testing.assert_close(dg, do, rtol=0, atol=2e-5, msg=lambda m: path + ": " + m)


@pytest.mark.parametrize("num_kf", [23, 33])
def test_ba_single_launch_cholesky_limits(db, O, dev, num_kf):
    """6P = 132 (not a panel multiple) and 6P = 192 (the LDS-resident path's cap: 151 KB of LDS)."""
    prob = _ba_problem(O, num_kf, 5 * num_kf, "tiny", seed=41)
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-4, 0.1, False)
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=2e-3, atol=5e-6)
    torch.testing.assert_close(pg, po, rtol=0, atol=2e-5)
    torch.testing.assert_close(dg, do, rtol=0, atol=2e-5)


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_altcorr_forward_matches_oracle(db, O, dev, dtype):
    """droid_backends.altcorr_forward vs the restatement of altcorr_kernel.cu:27-149: windows
    straddling every border, a pooled (smaller) fmap2, S = 2 coordinate sets."""
    g = torch.Generator().manual_seed(41)
    B, H1, W1, H2, W2, C, S = 3, 9, 13, 5, 7, 128, 2
    f1 = (torch.randn(B, H1, W1, C, generator=g) / 4).to(dtype)
    f2 = (torch.randn(B, H2, W2, C, generator=g) / 4).to(dtype)
    ys, xs = torch.meshgrid(torch.arange(H1, dtype=torch.float32), torch.arange(W1, dtype=torch.float32), indexing="ij")
    base = torch.stack([xs * (W2 / W1), ys * (H2 / H1)], -1)
    coords = base[None, None] + 2.0 * torch.randn(B, S, H1, W1, 2, generator=g)
    coords[:, :, 0, 0] = torch.tensor([-9.0, 2.0])
    coords[:, :, 0, 1] = torch.tensor([3.0, 2.0])
    ref, = O.altcorr_forward(f1.float(), f2.float(), coords, 3)
    out, = db.altcorr_forward(f1.to(dev), f2.to(dev), coords.to(dev), 3)
    assert out.dtype == dtype and tuple(out.shape) == (B, S, 49, H1, W1)
    if dtype == torch.float32:
        torch.testing.assert_close(out.cpu(), ref, rtol=1e-4, atol=1e-5)
    else:   # fp16 inputs, fp32 accumulation, fp16 output
        torch.testing.assert_close(out.cpu().float(), ref, rtol=2e-3, atol=2e-3)


def test_altcorr_block_matches_oracle_lookup(O, dev, built_lib):
    """AltCorrBlock(fmaps)(coords, ii, jj) == corr.py:95-145 on the oracle (4 levels, 196 channels)."""
    from go_slam_amd.corr import AltCorrBlock
    ht, wd, _ = synth.SHAPES["Scan"]
    fm = synth.make_features(5, "Scan", seed=43)[None]                 # [1,5,128,h,w] f16
    ii = torch.tensor([0, 1, 2, 4, 3, 1])
    jj = torch.tensor([1, 0, 4, 2, 3, 3])
    coords = _rand_coords(6, ht, wd, ht, wd, seed=44, spread=3.0).permute(0, 2, 3, 1)[None].contiguous()
    ref = O.altcorr_lookup(O.altcorr_pyramid(fm), coords, ii, jj, 3)
    blk = AltCorrBlock(fm.to(dev))
    out = blk(coords.to(dev), ii.to(dev), jj.to(dev))
    assert tuple(out.shape) == (1, 6, 196, ht, wd) and out.dtype == torch.float32
    torch.testing.assert_close(out.cpu(), ref, rtol=1e-3, atol=2e-3)


def _smooth_coords(n, ht, wd, seed, stretch=0.15, shift=4.0):
    """a reprojection-like flow: identity grid + per-edge shift + affine stretch + a low-frequency wobble"""
    g = torch.Generator().manual_seed(seed)
    ys, xs = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32), indexing="ij")
    out = []
    for _ in range(n):
        a = 1.0 + stretch * (torch.rand(4, generator=g) - 0.5)
        t = shift * torch.randn(2, generator=g)
        x = a[0] * xs + 0.05 * (a[1] - 1) * ys + t[0] + 0.7 * torch.sin(ys / 5.0 + t[1])
        y = a[2] * ys + 0.05 * (a[3] - 1) * xs + t[1] + 0.7 * torch.cos(xs / 6.0 + t[0])
        out.append(torch.stack([x, y], -1))
    return torch.stack(out)[None].contiguous()


@pytest.mark.parametrize("case", ["smooth flow (matrix-core path)", "3 px of per-pixel noise (both paths)",
                                  "ragged map 10x14 (partial tiles)", "wild coordinates (per-pixel path, nan / inf / 1e9)"])
def test_altcorr_pyramid_one_launch_matches_oracle_and_the_per_level_kernel(O, dev, built_lib, case):
    """AltCorrBlock.lookup = gs_altcorr_pyramid (all four levels, features indexed by ii / jj in the kernel, fp16
    channels-last output) vs the oracle's restatement of corr.py:112-145 + altcorr_kernel.cu:27-149, and vs the per-level
    gs_altcorr_forward path it replaces in update_lowmem (same fp16 features: at most one fp16 ulp apart)."""
    from go_slam_amd.corr import AltCorrBlock
    ht, wd = (10, 14) if case.startswith("ragged") else synth.SHAPES["Scan"][:2]
    g = torch.Generator().manual_seed(143)
    fm = (torch.randn(1, 5, 128, ht, wd, generator=g) * 1.5).half()
    ii = torch.tensor([0, 1, 2, 4, 3, 1, 0, 2, 4])
    jj = torch.tensor([1, 0, 4, 2, 3, 3, 4, 0, 1])              # nine edges: two rounds of the kernel's 8-edge XCD groups
    if case.startswith("smooth") or case.startswith("ragged"):
        coords = _smooth_coords(9, ht, wd, seed=144)
    else:
        coords = _rand_coords(9, ht, wd, ht, wd, seed=44, spread=3.0).permute(0, 2, 3, 1)[None].contiguous()
    if case.startswith("wild"):
        coords = coords * 3.0 - 20.0
        coords[0, 0, 2, 3] = torch.tensor([1.0e9, -1.0e9])
        coords[0, 1, 5, 5] = torch.tensor([float("inf"), 2.0])
        coords[0, 2, 7, 1] = torch.tensor([float("nan"), 3.0])
    ref = O.altcorr_lookup(O.altcorr_pyramid(fm), coords, ii, jj, 3)
    blk = AltCorrBlock(fm.to(dev))
    assert blk.fused_supported(coords.to(dev))
    out = blk.lookup(coords.to(dev), ii.to(dev), jj.to(dev))
    assert tuple(out.shape) == (1, 9, 196, ht, wd) and out.dtype == torch.float16
    assert out[0].is_contiguous(memory_format=torch.channels_last)
    per_level = blk(coords.to(dev), ii.to(dev), jj.to(dev))
    torch.cuda.synchronize()
    o, pl = out.float().cpu(), per_level.cpu()
    fin = torch.isfinite(ref)
    assert torch.equal(torch.isfinite(o), fin) and torch.equal(torch.isfinite(pl), fin)
    assert bool(fin.float().mean() > 0.99)
    ulp = 2.0 ** -10 * ref.abs().clamp(min=1.0)
    assert bool(((o - pl).abs()[fin] <= 1.01 * ulp[fin]).all()), float((o - pl).abs()[fin].max())
    torch.testing.assert_close(o[fin], ref[fin], rtol=1e-3, atol=2e-3)


@pytest.mark.parametrize("C", [64, 128])
def test_altcorr_backward_matches_oracle(db, O, dev, C):
    """altcorr_backward (training path, fp32) vs autograd over the oracle's forward restatement; target
    map smaller than the source map (a pooled pyramid level), windows hanging over the borders."""
    g = torch.Generator().manual_seed(81)
    B, S, H1, W1, H2, W2 = 2, 3, 6, 9, 5, 7
    f1 = torch.randn(B, H1, W1, C, generator=g)
    f2 = torch.randn(B, H2, W2, C, generator=g)
    coords = torch.stack([torch.rand(B, S, H1, W1, generator=g) * (W2 + 4) - 2,
                          torch.rand(B, S, H1, W1, generator=g) * (H2 + 4) - 2], -1).contiguous()
    cg = torch.randn(B, S, 49, H1, W1, generator=g)
    ref = O.altcorr_backward(f1, f2, coords, cg, 3)
    out = db.altcorr_backward(f1.to(dev), f2.to(dev), coords.to(dev), cg.to(dev), 3)
    assert len(out) == 3 and not bool(out[2].any())                     # coords_grad: zeros, as the reference
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=1e-4, atol=1e-4)
    torch.testing.assert_close(out[1].cpu(), ref[1], rtol=1e-4, atol=1e-4)


def test_ba_global_window_200_keyframes(db, O, dev):
    """BASELINE configs[3] stress shape: >= 200 keyframes in one dense BA (6P = 1254 unknowns, 1000 edges):
    multi-kernel blocked Cholesky, long CSR rows, Schur pair lists."""
    prob = _ba_problem(O, 210, 1000, "tiny", seed=83)
    ref, (po, do), out, (pg, dg) = _run_ba_pair(db, O, dev, prob, 2, 1e-5, 1e-2, False)
    torch.testing.assert_close(out[0].cpu(), ref[0], rtol=5e-3, atol=1e-5)
    torch.testing.assert_close(pg, po, rtol=0, atol=5e-5)
    torch.testing.assert_close(dg, do, rtol=0, atol=5e-5)


@pytest.mark.parametrize("shape", ["tiny", "Scan", "Rep"])
def test_corr_volume_pyramid_matches_oracle(db, O, dev, shape):
    """Fused MFMA volume + pyramid vs CorrBlock.corr/avg_pool2d restated on the CPU.  Level 0 may
    differ by one fp16 rounding of the fp32 dot product (summation order); pooled levels must be
    EXACTLY the average-pool of the level below (checked against the GPU's own lower level)."""
    import torch.nn.functional as F
    ht, wd, _ = synth.SHAPES[shape]
    n = 2
    f1 = synth.make_features(n, shape, seed=51)
    f2 = synth.make_features(n, shape, seed=52)
    ref = O.corr_pyramid(f1[None], f2[None])
    out = db.corr_volume_pyramid(f1.to(dev), f2.to(dev))
    for l in range(4):
        assert tuple(out[l].shape) == tuple(ref[l].shape), (l, out[l].shape, ref[l].shape)
    o0, r0 = out[0].cpu().float(), ref[0].float()
    diff = (o0 - r0).abs()
    assert float(diff.max()) <= 2 ** -10 * max(1.0, float(r0.abs().max())) * 1.01
    assert float((diff == 0).float().mean()) > 0.97
    for l in range(1, 4):
        low = out[l - 1].cpu()
        nn_, h1, w1, hl, wl = low.shape
        pooled = F.avg_pool2d(low.reshape(-1, 1, hl, wl).float(), 2, 2).to(torch.float16)
        assert torch.equal(out[l].cpu().reshape(-1, 1, hl // 2, wl // 2), pooled), f"level {l}"
        torch.testing.assert_close(out[l].cpu().float(), ref[l].float(), rtol=0, atol=2e-3)


@pytest.mark.parametrize("shape", ["tiny", "Rep", "S480"])
def test_corr_tile8_layout_is_a_pure_relayout(db, dev, shape):
    """The tile8 volume layout (levels 0-1 in 128-byte tiles) must hold exactly the row-major values, and
    the lookup through it must be bit-identical, including windows hanging over every border."""
    ht, wd, _ = synth.SHAPES[shape]
    n = 2 if shape != "S480" else 1
    f1 = synth.make_features(n, shape, seed=71).to(dev)
    f2 = synth.make_features(n, shape, seed=72).to(dev)
    rm = db.corr_volume_pyramid(f1, f2)
    t8 = db.corr_volume_pyramid(f1, f2, layout=db.CORR_TILE8)
    for l in range(2):
        assert torch.equal(db.corr_untile8(t8[l], ht >> l, wd >> l), rm[l]), f"level {l}"
    for l in (2, 3):
        assert torch.equal(t8[l], rm[l])
    for spread in (3.0, 40.0):               # 40 px: many windows partly or wholly outside the map
        coords = _rand_coords(n, ht, wd, ht, wd, seed=73, spread=spread).permute(0, 2, 3, 1).contiguous().to(dev)
        a = db.corr_lookup_pyramid(rm, coords, 3, channels_last=True)
        b = db.corr_lookup_pyramid(t8, coords, 3, channels_last=True, layout=db.CORR_TILE8, map_size=(ht, wd))
        assert torch.equal(a, b), f"spread {spread}"


def test_quirk_q1_first_window_pose_never_feeds_back_into_dz(db, O, dev):
    """SURVEY App. A Q1 (droid_kernels.cu:1105): entries whose pose index - t0 <= 0 are skipped in
    dz = Q (w - E^T dx).  The HIP path must match the oracle WITH the quirk and differ from a
    'corrected' back-substitution."""
    prob = _ba_problem(O, 8, 22, "tiny", seed=61)
    K = prob["intrinsics"][0].contiguous()
    args = (K, prob["disps_sens"], prob["target"], prob["weight"], prob["eta"], prob["ii"], prob["jj"],
            prob["t0"], prob["t1"], 1, 1e-4, 0.1, False)
    with_q = O.ba(prob["poses"].clone(), prob["disps"].clone(), *args)
    without_q = O.ba(prob["poses"].clone(), prob["disps"].clone(), *args, evt_quirk=False)
    _, _, out, _ = _run_ba_pair(db, O, dev, prob, 1, 1e-4, 0.1, False)
    torch.testing.assert_close(out[1].cpu(), with_q[1], rtol=1e-3, atol=1e-5)
    assert (with_q[1] - without_q[1]).abs().max() > 50 * (out[1].cpu() - with_q[1]).abs().max()


def test_quirk_q2_two_min_depths(db, O, dev):
    """Q2: the kernels use MIN_DEPTH 0.25 (droid_kernels.cu:26), the Python reprojection 0.2
    (projective_ops.py:4).  A point that lands at z = 0.22 is valid for reproject, invalid for projmap."""
    ht, wd = 12, 16
    vid = synth.make_video(2, "tiny", seed=63)
    vid["poses"][:] = torch.tensor([0.0, 0, 0, 0, 0, 0, 1])
    vid["poses"][1, 2] = -0.78                       # frame 1 sits 0.78 m ahead along z
    vid["disps"][:] = 1.0                            # all points at depth 1 => z in frame 1 = 0.22
    ii, jj = torch.tensor([0]), torch.tensor([1])
    K = vid["intrinsics"][0].contiguous()
    c, v = db.reproject(vid["poses"].to(dev), vid["disps"].to(dev), vid["intrinsics"].to(dev), ii.to(dev), jj.to(dev))
    pc, pv = db.projmap(vid["poses"].to(dev), vid["disps"].to(dev), K.to(dev), ii.to(dev), jj.to(dev))
    assert v.min() == 1.0 and pv.max() == 0.0
    rc, rv = O.reproject(vid["poses"], vid["disps"], vid["intrinsics"], ii, jj)
    assert torch.equal(v.cpu(), rv)


def test_empty_inputs_are_noops(db, dev):
    """Zero edges / frames / rays: every entry point returns correctly shaped empty results instead of launching
    (the reference's kernels would be launched with a zero-sized grid, which CUDA rejects)."""
    h, w = 12, 16
    vid = synth.make_video(4, "tiny", seed=91)
    poses, disps, K = vid["poses"].to(dev), vid["disps"].to(dev), vid["intrinsics"].to(dev)
    e = torch.zeros(0, dtype=torch.long, device=dev)
    c, v = db.reproject(poses, disps, K, e, e)
    assert tuple(c.shape) == (1, 0, h, w, 2) and tuple(v.shape) == (1, 0, h, w, 1)
    assert tuple(db.frame_distance(poses, disps, K[0].contiguous(), e, e, 0.3).shape) == (0,)
    pm = db.projmap(poses, disps, K[0].contiguous(), e, e)
    assert pm[0].shape[0] == 0
    vol = torch.zeros(0, h, w, h, w, device=dev, dtype=torch.float16)
    out, = db.corr_index_forward(vol, torch.zeros(0, 2, h, w, device=dev), 3)
    assert tuple(out.shape) == (0, 7, 7, h, w)
    pyr = [torch.zeros(0, h, w, h >> l, w >> l, device=dev, dtype=torch.float16) for l in range(4)]
    out = db.corr_lookup_pyramid(pyr, torch.zeros(0, h, w, 2, device=dev), 3, channels_last=True)
    assert tuple(out.shape) == (0, 196, h, w)
    f = torch.zeros(0, h, w, 128, device=dev)
    out, = db.altcorr_forward(f, f, torch.zeros(0, 1, h, w, 2, device=dev), 3)
    assert out.shape[0] == 0


def test_oversized_batches_raise(db, dev):
    """Launch-geometry limits are reported as errors, never silently truncated."""
    n, h, w = 65536, 2, 2
    vol = torch.zeros(n, h, w, h, w, device=dev, dtype=torch.float16)
    with pytest.raises(RuntimeError, match="exceeds"):
        db.corr_index_forward(vol, torch.zeros(n, 2, h, w, device=dev), 3)


@pytest.mark.parametrize("hw,expect_fused", [((16, 32), True), ((16, 96), True), ((12, 20), True), ((8, 104), False),
                                             ((12, 18), False)])
def test_corrblock_matches_oracle_on_fused_and_fallback_shapes(O, dev, hw, expect_fused):
    """CorrBlock (build + 4-level lookup) vs the oracle at shapes the fused MFMA builder covers (w % 8 == 4 included since
    round 5) and at two it does not (width > 96, width not a multiple of 4: library GEMM + avg_pool2d on the GPU, announced
    by a RuntimeWarning, then the same HIP lookup)."""
    from go_slam_amd import droid_backends as db
    from go_slam_amd.corr import CorrBlock
    h, w = hw
    g = torch.Generator().manual_seed(95)
    f1 = torch.randn(1, 2, 128, h, w, generator=g).half()
    f2 = torch.randn(1, 2, 128, h, w, generator=g).half()
    assert db.corr_volume_supported(f1[0]) == expect_fused
    coords = _rand_coords(2, h, w, h, w, seed=96, spread=3.0).permute(0, 2, 3, 1)[None].contiguous()
    ref = O.corr_lookup(O.corr_pyramid(f1, f2), coords, 3)
    for cl in (False, True):
        blk = CorrBlock(f1.to(dev), f2.to(dev), channels_last=cl)
        out = blk(coords.to(dev))
        assert tuple(out.shape) == (1, 2, 196, h, w)
        # level 0 may differ by one fp16 rounding of the fp32 dot product between GEMM implementations
        torch.testing.assert_close(out.cpu().float(), ref.float(), rtol=0, atol=4e-3)


@pytest.mark.parametrize("h,w,layout", [(16, 96, "tile8"), (16, 96, "rowmajor"), (12, 88, "rowmajor"),
                                        (40, 60, "rowmajor"), (10, 92, "rowmajor"), (9, 12, "rowmajor")])
def test_corr_volume_pyramid_at_the_kernel_width_limit(db, O, dev, h, w, layout):
    """corr_volume_kernel's real limit is three 32-column tiles per wave, w <= 96 (the Python gate said 80 until round
    4): level 0 within one fp16 rounding of the oracle's fp32 dot products, pooled levels exactly the pool of the level
    below, in both layouts (tile8 needs w % 16 == 0).  Round 5: widths with w % 8 == 4 -- EuRoC's 320 x 480 input = 40 x 60
    maps (configs/EuRoC/mh_01_easy.yaml), 92, 12 -- where a block of four target rows ends in a half-full MFMA column tile,
    and map heights that are not a multiple of the block (10, 9)."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(500 + w)
    f1 = torch.randn(2, 128, h, w, generator=g).half()
    f2 = torch.randn(2, 128, h, w, generator=g).half()
    assert db.corr_volume_supported(f1) and db.CORR_VOLUME_MAX_W == 96
    ref = O.corr_pyramid(f1[None], f2[None])
    lay = db.CORR_TILE8 if layout == "tile8" else db.CORR_ROWMAJOR
    out = db.corr_volume_pyramid(f1.to(dev), f2.to(dev), layout=lay)
    if lay == db.CORR_TILE8:
        out = [db.corr_untile8(out[l], h >> l, w >> l) if l < 2 else out[l] for l in range(4)]
    o0, r0 = out[0].cpu().float(), ref[0].float().reshape(out[0].shape)
    diff = (o0 - r0).abs()
    assert float(diff.max()) <= 2 ** -10 * max(1.0, float(r0.abs().max())) * 1.01
    for l in range(1, 4):
        low = out[l - 1].cpu()
        hl, wl = low.shape[-2:]
        pooled = F.avg_pool2d(low.reshape(-1, 1, hl, wl).float(), 2, 2).to(torch.float16)
        assert torch.equal(out[l].cpu().reshape(-1, 1, hl // 2, wl // 2), pooled), f"level {l}"
    assert db.corr_volume_supported(torch.zeros(1, 128, 40, 60).half())             # EuRoC: w % 8 == 4
    assert not db.corr_volume_supported(torch.zeros(1, 128, 40, 62).half())         # w % 4 != 0
    assert not db.corr_volume_supported(torch.zeros(1, 128, 16, 104).half())


def test_corr_block_at_the_euroc_map_size_runs_the_own_kernels(db, O, dev):
    """EuRoC's 320 x 480 input gives 40 x 60 maps (configs/EuRoC/mh_01_easy.yaml, v1_02_medium, v2_01_easy): w % 8 == 4.
    Until round 5 CorrBlock took the reference's matmul + avg_pool2d formulation there (with a RuntimeWarning); now the
    volume comes from gs_corr_volume_pyramid -- no warning, no library GEMM -- and the lookup equals the oracle's."""
    import warnings
    from go_slam_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(611)
    f1 = torch.randn(1, 2, 128, 40, 60, generator=g).half()
    f2 = torch.randn(1, 2, 128, 40, 60, generator=g).half()
    CorrBlock._warned_fallback = False
    calls = []
    orig = torch.matmul
    torch.matmul = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            blk = CorrBlock(f1.to(dev), f2.to(dev))
    finally:
        torch.matmul = orig
    assert not calls and not any(issubclass(w_.category, RuntimeWarning) for w_ in rec)
    ys, xs = torch.meshgrid(torch.arange(40, dtype=torch.float32), torch.arange(60, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, 2, 40, 60, 2, generator=g)).contiguous()
    out = blk(coords.to(dev))
    ref = O.corr_lookup(O.corr_pyramid(f1, f2), coords, 3)
    torch.testing.assert_close(out.float().cpu().reshape(ref.shape), ref.float(), rtol=0, atol=2e-2)
    # the frontend's form: channels-last features (row-major volume at this width: tile8 needs w % 16 == 0)
    blk_cl = CorrBlock(f1.to(dev), f2.to(dev), channels_last=True)
    assert blk_cl.layout == db.CORR_ROWMAJOR
    out_cl = blk_cl(coords.to(dev))
    assert torch.equal(out_cl.float().cpu().reshape(ref.shape), out.float().cpu().reshape(ref.shape))


def test_corr_block_outside_the_kernel_shapes_warns_instead_of_silently_using_the_library(db, O, dev):
    """Widths the fused builder does not cover (w % 4 != 0; here 20 x 30 maps, w % 4 == 2): CorrBlock takes the reference's own
    formulation (matmul + avg_pool2d) -- with a RuntimeWarning, once -- and its lookup still equals the oracle's."""
    import warnings
    from go_slam_amd.corr import CorrBlock
    g = torch.Generator().manual_seed(612)
    f1 = torch.randn(1, 2, 128, 20, 30, generator=g).half()
    f2 = torch.randn(1, 2, 128, 20, 30, generator=g).half()
    CorrBlock._warned_fallback = False
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        blk = CorrBlock(f1.to(dev), f2.to(dev))
        CorrBlock(f1.to(dev), f2.to(dev))
    assert sum(issubclass(w_.category, RuntimeWarning) and "gs_corr_volume_pyramid" in str(w_.message) for w_ in rec) == 1
    ys, xs = torch.meshgrid(torch.arange(20, dtype=torch.float32), torch.arange(30, dtype=torch.float32), indexing="ij")
    coords = (torch.stack([xs, ys], -1)[None, None] + 2.0 * torch.randn(1, 2, 20, 30, 2, generator=g)).contiguous()
    out = blk(coords.to(dev))
    ref = O.corr_lookup(O.corr_pyramid(f1, f2), coords, 3)
    torch.testing.assert_close(out.float().cpu().reshape(ref.shape), ref.float(), rtol=0, atol=2e-2)


def test_hip_kernels_match_the_reference_kernel_fixture(db, O, dev):
    """The HIP kernels against the REFERENCE's own kernels, no restatement in between: tests/golden/reference_kernels.npz holds
    what src/lib/*.cu produced (compiled for the CPU by oracle/build_ref.py, run in the build container:
    gen_golden.py::gen_reference_kernels) for the seeded inputs below -- two `ba` iterations on the 12-keyframe / 40-edge ScanNet-shaped window of
    test_ba_two_iterations_frontend_like, and the geometry kernels on the video of
    test_projmap_frame_distance_iproj_depth_filter."""
    g = _fixture("reference_kernels.npz")
    prob = _ba_problem(O, 12, 40, "Scan", seed=13)
    K = prob["intrinsics"][0].contiguous()
    pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
    out = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), prob["t0"], prob["t1"], 2, 1e-4, 0.1, False)
    torch.testing.assert_close(out[0].cpu(), g["ba_dx"], rtol=1e-3, atol=3.5e-6)
    torch.testing.assert_close(pg.cpu(), g["ba_poses"], rtol=0, atol=1.2e-5)
    torch.testing.assert_close(dg.cpu(), g["ba_disps"], rtol=0, atol=1.2e-5)
    vid = synth.make_video(10, "tiny", seed=9)
    ii, jj = synth.make_graph(10, 30, seed=9)
    Pd, Dd, Kd = vid["poses"].to(dev), vid["disps"].to(dev), vid["intrinsics"][0].contiguous().to(dev)
    c, v = db.projmap(Pd, Dd, Kd, ii.to(dev), jj.to(dev))
    assert torch.equal(v.cpu(), g["projmap_valid"])
    torch.testing.assert_close(c.cpu(), g["projmap_coords"], rtol=1e-6, atol=1e-4)
    for beta, key in ((0.3, "frame_distance_03"), (0.7, "frame_distance_07")):
        torch.testing.assert_close(db.frame_distance(Pd, Dd, Kd, ii.to(dev), jj.to(dev), beta).cpu(), g[key],
                                   rtol=1.1e-4, atol=1.2e-5)
    torch.testing.assert_close(db.iproj(Pd, Dd, Kd).cpu(), g["iproj"], rtol=1e-6, atol=1e-5)
    out = db.depth_filter(Pd, Dd, Kd, torch.tensor([0, 1, 4, 8, 9]).to(dev), torch.tensor([0.05, 0.1, 0.2, 0.05, 0.3]).to(dev))
    assert torch.equal(out.cpu(), g["depth_filter"])
    # lookups (correlation_kernels.cu:19-70) on the inputs of test_corr_index_forward_matches_oracle
    coords = _rand_coords(3, 12, 16, 12, 16)
    for tag, dt in (("f16", torch.float16), ("f32", torch.float32)):
        vol = _rand_volume(3, 12, 16, 12, 16, dt)
        out, = db.corr_index_forward(vol.to(dev), coords.to(dev), 3)
        if dt == torch.float16:
            assert torch.equal(out.float().cpu(), g["lookup_" + tag])
        else:
            torch.testing.assert_close(out.cpu(), g["lookup_" + tag], rtol=1e-5, atol=1e-6)
    # one `ba` iteration with and without the depth prior, on the problem of test_ba_one_iteration_matches_oracle
    for tag, rgbd in (("rgbd", True), ("mono", False)):
        prob = _ba_problem(O, 8, 22, "tiny", seed=11, rgbd=rgbd)
        K = prob["intrinsics"][0].contiguous()
        pg, dg = prob["poses"].clone().to(dev), prob["disps"].clone().to(dev)
        out = db.ba(pg, dg, K.to(dev), prob["disps_sens"].to(dev), prob["target"].to(dev), prob["weight"].to(dev),
                    prob["eta"].to(dev), prob["ii"].to(dev), prob["jj"].to(dev), prob["t0"], prob["t1"], 1, 1e-4, 0.1,
                    False)
        torch.testing.assert_close(out[0].cpu(), g[f"ba1_{tag}_dx"], rtol=2.1e-4, atol=2.5e-6)
        torch.testing.assert_close(out[1].cpu(), g[f"ba1_{tag}_dz"], rtol=1.1e-3, atol=1.3e-5)
        torch.testing.assert_close(pg.cpu(), g[f"ba1_{tag}_poses"], rtol=0, atol=1.3e-5)
        torch.testing.assert_close(dg.cpu(), g[f"ba1_{tag}_disps"], rtol=0, atol=1.3e-5)



@pytest.mark.parametrize("shape", ["Scan", "S480"])
def test_corr_pool_equals_corr_block_through_adds_removals_and_growth(db, dev, built_lib, shape):
    """corr.CorrPool (the factor graph's volume store: edges own SLOTS of a capacity buffer, gs_corr_volume_pyramid_slots
    builds in place, the lookups take the slot list) against CorrBlock with the reference's cat / [keep] copies
    (src/factor_graph.py:118,150), through the life of a graph: append, drop by mask (prefix and scattered), append into
    reused slots, growth past the capacity, drop by index list.  Planes, the materialised lookup and the lookup fused with
    corr_encoder[0] must be IDENTICAL bit for bit -- same kernels, one more index."""
    from go_slam_amd.corr import CorrBlock, CorrPool
    ht, wd, _ = synth.SHAPES[shape]
    g = torch.Generator().manual_seed(301)
    fm = lambda n: torch.randn(1, n, 128, ht, wd, generator=g).half().to(dev)
    pool = CorrPool(ht, wd, dev, capacity=6)
    blk = None
    assert CorrPool.supported(fm(1))

    def add(n):
        nonlocal blk
        f1, f2 = fm(n), fm(n)
        pool.append(f1, f2)
        b = CorrBlock(f1, f2, channels_last=True)
        blk = b if blk is None else blk.cat(b)

    def drop(index):
        nonlocal blk
        pool[index]
        blk = blk[index]

    def check():
        E = len(pool)
        assert E == blk.corr_pyramid[0].shape[0] and pool.layout == blk.layout
        assert int(pool.free.numel()) + E == pool.capacity and len(set(pool.slot.tolist())) == E
        for l, (a, b) in enumerate(zip(pool.corr_pyramid, blk.corr_pyramid)):
            if pool.layout == db.CORR_TILE8 and l < 2:      # tile8 planes are padded to whole 8 x 8 tiles: the padding rows
                hl, wl = ht >> l, wd >> l                   # are never written (nor read) -- compare what a lookup can see
                a, b = db.corr_untile8(a, hl, wl)[..., :hl, :wl], db.corr_untile8(b, hl, wl)[..., :hl, :wl]
            assert torch.equal(a.reshape(b.shape), b), l
        if E == 0:
            return
        ys, xs = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32), indexing="ij")
        coords = (torch.stack([xs, ys], -1)[None, None] + 4.0 * torch.randn(1, E, ht, wd, 2, generator=g)).to(dev)
        assert torch.equal(pool(coords), blk(coords))
        wpad = torch.zeros(128, 208, dtype=torch.float16, device=dev)
        wpad[:, :196] = (torch.randn(128, 196, generator=g) / 14.0).half().to(dev)
        bias = torch.randn(128, generator=g).to(dev)
        assert torch.equal(pool.lookup_encoded(coords, wpad, bias), blk.lookup_encoded(coords, wpad, bias))

    add(4)
    check()
    drop(torch.tensor([False, False, True, True], device=dev))                 # retire the oldest two (a prefix)
    add(3)                                                                     # lands in the freed slots first
    check()
    assert pool.capacity == 6
    drop(torch.tensor([True, False, True, False, True], device=dev))           # scattered (max_factors retirement)
    add(7)                                                                     # 3 + 7 > 6: the buffers grow, live slots keep their numbers
    assert pool.capacity >= 10
    check()
    drop(torch.tensor([9, 0, 4, 5], device=dev))                               # an index list, reordering the edges
    check()
    drop(torch.zeros(4, dtype=torch.bool, device=dev))                         # clear_edges
    check()
    add(2)
    check()
