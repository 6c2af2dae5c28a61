"""Per-frame host pieces of the tracker on the CPU: the encoders, the DroidNet checkpoint contract, MotionFilter and
the DepthVideo item setter -- against fixtures produced by the reference's own modules (tests/golden/gen_golden.py:
gen_encoder, gen_motion_filter)."""
import importlib.util
import os
import types

import numpy as np
import pytest
import torch

from go_slam_amd.depth_video import DepthVideo
from go_slam_amd.droid_net import DroidNet, load_pretrained

HERE = os.path.dirname(__file__)


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "golden", "gen_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


@pytest.fixture(scope="module")
def net():
    gen = _gen()
    n = DroidNet().eval()
    n.load_state_dict(gen.named_weights(n.state_dict(), seed=173))
    return n


def test_droidnet_checkpoint_contract():
    """same parameter names, order and shapes as the reference's DroidNet (strict load_state_dict, src/slam.py:196-208);
    load_pretrained strips `module.` and slices the 3-channel heads of the published checkpoint to 2."""
    gold = np.load(os.path.join(HERE, "golden", "encoders.npz"))
    n = DroidNet()
    sd = n.state_dict()
    assert list(sd.keys()) == list(gold["keys"])
    assert [str(tuple(v.shape)) for v in sd.values()] == list(gold["shapes"])
    ckpt = {"module." + k: v.clone() for k, v in sd.items()}
    for head in ("weight", "delta"):
        for p in ("weight", "bias"):
            k = f"module.update.{head}.2.{p}"
            ckpt[k] = torch.cat([ckpt[k], torch.full_like(ckpt[k][:1], 7.0)])       # 3 output channels, as trained
    m = load_pretrained(DroidNet(), ckpt)
    for k, v in m.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_encoders_match_reference(net):
    """BasicEncoder in both DroidNet configurations vs the reference's (src/modules/extractor.py:61-126), CPU fp32"""
    gold = np.load(os.path.join(HERE, "golden", "encoders.npz"))
    x = torch.from_numpy(gold["x"])
    with torch.no_grad():
        f, c = net.fnet(x), net.cnet(x)
    assert f.shape == (1, 2, 128, 8, 12) and c.shape == (1, 2, 256, 8, 12)
    assert np.allclose(f.numpy(), gold["fnet"], rtol=1e-4, atol=1e-4)
    assert np.allclose(c.numpy(), gold["cnet"], rtol=1e-4, atol=1e-4)


class _ZeroCorr:
    """the HIP CorrBlock cannot run on the CPU; the scripted update operator ignores the features anyway"""

    def __init__(self, fmap1, fmap2):
        assert fmap1.shape == fmap2.shape and fmap1.shape[:3] == (1, 1, 128)

    def __call__(self, coords):
        return torch.zeros(1, 1, 196, *coords.shape[2:4])


@pytest.mark.parametrize("stereo", [False, True])
def test_motion_filter_matches_reference(net, stereo, monkeypatch):
    """keyframe decisions and every argument of video.append vs the reference's MotionFilter.track
    (src/motion_filter.py:41-90): first frame with identity pose and unit disparity, later keyframes with None,
    intrinsics / 8, left-view net / inp, both views' feature maps, the in-place normalised image alias."""
    import go_slam_amd.motion_filter as MF
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "motion_filter.npz"))
    monkeypatch.setattr(MF, "CorrBlock", _ZeroCorr)
    tag = "stereo" if stereo else "mono"
    for value_counter in (True, False):
        appended, counts = gen.run_motion_filter(MF.MotionFilter, net, stereo, value_counter)
        assert counts == list(gold[f"{tag}_counts"])
        assert len(appended) == int(gold[f"{tag}_n"])
        for k, item in enumerate(appended):
            assert [x is None for x in item] == list(gold[f"{tag}_{k}_none"]), k
            for a, x in enumerate(item):
                if x is not None:
                    assert np.allclose(gen.digest(x).numpy(), gold[f"{tag}_{k}_{a}"], rtol=1e-4, atol=1e-4), (k, a)


def test_depth_video_item_setter_and_normalize():
    """DepthVideo.append / __setitem__ / __getitem__ / normalize (src/depth_video.py:80-143, 198-205)"""
    v = DepthVideo(4, 6, buffer=8, device="cpu", stereo=True, full_res=True)
    g = torch.Generator().manual_seed(1)
    depth = torch.rand(32, 48, generator=g) * 3
    depth[::5] = 0.0                                                  # missing sensor depth stays 0
    img = torch.rand(3, 32, 48, generator=g)
    fmap = torch.rand(2, 128, 4, 6, generator=g)
    net_, inp_ = torch.rand(128, 4, 6, generator=g), torch.rand(128, 4, 6, generator=g)
    intr = torch.tensor([10.0, 11.0, 3.0, 2.0])
    ident = torch.tensor([0, 0, 0, 0, 0, 0, 1.0])
    v.append(0.5, img, ident, 1.0, depth, intr, fmap, net_, inp_, torch.eye(4) * 2)
    assert v.counter.value == 1 and float(v.timestamp[0]) == 0.5
    sub = depth[3::8, 3::8]
    want = torch.where(sub > 0, 1.0 / sub, sub)
    assert torch.equal(v.disps_sens[0], want) and torch.equal(v.disps[0], want)      # sensor depth overrides 1.0
    assert torch.equal(v.images[0], img) and torch.equal(v.depths_gt[0], depth)
    assert torch.equal(v.fmaps[0], fmap.half()) and torch.equal(v.nets[0], net_.half())
    assert torch.equal(v.poses_gt[0], torch.eye(4) * 2)
    v.append(1.5, img, None, None, None, intr, fmap, net_, inp_, None)
    assert v.counter.value == 2 and torch.equal(v.poses[1], ident) and torch.all(v.disps[1] == 1.0)
    v[5] = (9.0, img, None, 0.25, None, None)                         # sparse write moves the counter
    assert v.counter.value == 6 and torch.all(v.disps[5] == 0.25)
    poses, disps, intrinsics, fmaps, nets, inps = v[0]
    assert torch.equal(disps, v.disps[0]) and torch.equal(intrinsics, intr)
    v.counter = 2
    v.poses[:2, :3] = torch.tensor([[1.0, 2.0, 3.0], [2.0, 0.0, 1.0]])
    before = v.disps[:2].clone()
    s = before.mean()
    v.normalize()
    assert torch.allclose(v.disps[:2], before / s) and torch.allclose(v.disps[:2].mean(), torch.tensor(1.0))
    assert torch.allclose(v.poses[0, :3], torch.tensor([1.0, 2.0, 3.0]) * s) and bool(v.dirty[:2].all())
    # slice write as PoseTrajectoryFiller does (src/trajectory_filler.py:64): the counter is NOT moved by it
    v.counter = 2
    tt = torch.tensor([7.0, 8.0])
    Gs = ident.repeat(2, 1)
    Gs[:, 0] = torch.tensor([0.5, 0.6])
    v[2:4] = (tt, torch.stack([img, img]), Gs, 1, torch.stack([depth, depth]), intr.repeat(2, 1) / 8.0,
              torch.stack([fmap, fmap]))
    assert v.counter.value == 2 and torch.equal(v.timestamp[2:4], tt) and torch.equal(v.poses[2:4], Gs)
    assert torch.equal(v.disps[2], want) and torch.equal(v.fmaps[3], fmap.half())
    lean = DepthVideo(4, 6, buffer=4, device="cpu")                   # hot-path-only mirror: no full-res buffers
    lean.append(0.0, img, ident, 1.0, depth, intr, fmap[:1], net_, inp_, None)
    assert lean.counter.value == 1 and not hasattr(lean, "images")
    cfg = {"cam": {"H_out": 32, "W_out": 48}, "tracking": {"buffer": 5}, "mode": "rgbd"}
    fc = DepthVideo.from_config(cfg, types.SimpleNamespace(device="cpu"))
    assert (fc.ht, fc.wd) == (32, 48) and (fc.map_ht, fc.map_wd) == (4, 6) and fc.images.shape == (5, 3, 32, 48) and not fc.stereo


def test_multiview_filter_matches_reference(monkeypatch):
    """MultiviewFilter.forward host logic vs the reference's (src/multiview_filter.py:99-173; fixture
    multiview_filter.npz): masks after dilation + strict in-bound test, filtered disparities / poses, pose-change
    priorities, scene bound, the two `< 100 points` early-outs and the warm-up gate.  The two native calls (iproj,
    depth_filter; parity-tested on the GPU in test_track_gpu.py) are stood in by the oracle on both sides, so this pins
    the device-resident masked-reduction formulation against the reference's host-side boolean indexing."""
    import go_slam_amd.multiview_filter as MV
    from oracle import droid_oracle as DO
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "multiview_filter.npz"))
    monkeypatch.setattr(MV, "droid_backends", types.SimpleNamespace(iproj=DO.iproj, depth_filter=DO.depth_filter))
    res = gen.run_filter_cases(MV.MultiviewFilter)
    assert set(res) == {c[0] for c in gen.MVF_CASES}
    for name, r in res.items():
        assert np.array_equal(r["mask_filtered"].numpy(), gold[f"{name}_mask_filtered"]), name
        assert np.array_equal(r["filtered_id"].numpy(), gold[f"{name}_filtered_id"]), name
        assert np.array_equal(r["disps_filtered"].numpy(), gold[f"{name}_disps_filtered"]), name
        assert np.array_equal(r["poses_filtered"].numpy(), gold[f"{name}_poses_filtered"]), name
        assert np.allclose(r["bound"].numpy(), gold[f"{name}_bound"], rtol=0, atol=1e-6), name
        assert np.allclose(r["update_priority"].numpy(), gold[f"{name}_update_priority"], rtol=1e-6, atol=1e-6), name
    assert int(gold["few_filtered_id"][0]) == -1 and int(gold["k3_filtered_id"][0]) == 12


def test_masked_bound_equals_compaction():
    from go_slam_amd.multiview_filter import in_bound, masked_bound
    g = torch.Generator().manual_seed(3)
    pts = torch.randn(5, 7, 9, 3, generator=g)
    m = torch.rand(5, 7, 9, generator=g) > 0.6
    sel = pts[m]
    want = torch.stack([sel.min(0).values, sel.max(0).values], -1)
    assert torch.equal(masked_bound(pts, m), want)
    grown = masked_bound(pts, m, enlarge_scale=1.5)
    assert torch.allclose(grown[:, 1] - grown[:, 0], 1.5 * (want[:, 1] - want[:, 0]))
    inside = in_bound(pts, want)
    assert inside.shape == m.shape and not bool(inside[m].all())       # extremal points sit ON the box: strict test
    assert int(inside[m].sum()) >= int(m.sum()) - 6


def test_trajectory_filler_matches_reference(net):
    """PoseTrajectoryFiller vs the reference's (src/trajectory_filler.py:29-112; fixture trajectory_filler.npz):
    keyframe bracketing incl. the `-1` bracket of a frame earlier than the first keyframe, constant-velocity SE3
    interpolation, the items parked behind the keyframes (timestamps, left image, poses, unit disparity, depth,
    intrinsics / 8, features), the two edge sets per batch, six motion-only updates over [N, N + M), batches of 16."""
    import go_slam_amd.trajectory_filler as TF
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "trajectory_filler.npz"))
    with torch.no_grad():
        poses, rec = gen.run_filler(TF.PoseTrajectoryFiller, net, TF)
    assert np.allclose(poses.numpy(), gold["poses"], atol=1e-5)
    assert repr(rec["factors"]) == str(gold["factors"])
    assert repr(rec["updates"]) == str(gold["updates"])
    assert rec["counter"] == list(gold["counter"]) and len(rec["set"]) == int(gold["n_set"])
    for k, (a, b, items) in enumerate(rec["set"]):
        assert [a, b] == list(gold[f"set{k}_range"])
        assert [x is None for x in items] == list(gold[f"set{k}_none"])
        for j, x in enumerate(items):
            if x is not None:
                assert np.allclose(x.numpy(), gold[f"set{k}_{j}"], rtol=1e-4, atol=1e-4), (k, j)


def test_mapper_schedule_and_ray_batches_match_reference():
    """Mapper.__call__ + DepthVideo.get_mapping_item vs the reference's Mapper running on the reference's own
    DepthVideo (src/mapping.py:151-302, src/depth_video.py:153-177; fixture mapper.npz): which keyframes are visited
    (new, two most recent, 10 highest priority, stratified-random old ones), the 10x first-call factor and the
    `the_end` factor, priority decay per hand-out (duplicates decay twice), the bound hand-over, and every ray batch
    (origins, directions, colours, depths) under the same seeds."""
    from go_slam_amd.neus.mapping import Mapper
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "mapper.npz"))
    video = DepthVideo.from_config(gen.mapper_cfg(), types.SimpleNamespace(device="cpu"))
    gen.fill_mapping_video(video)
    item = video.get_mapping_item(7, "cpu", decay=1.0)
    for k, x in enumerate(item):
        assert np.allclose(x.numpy(), gold[f"item7_{k}"], rtol=1e-5, atol=1e-6), k
    rec, mapper = gen.run_mapper(Mapper, video)
    struct, nums = gen.flatten_mapper_record(rec)
    assert struct == str(gold["struct"])
    assert len(nums) == int(gold["n"])
    for k, x in enumerate(nums):
        assert np.allclose(x.numpy(), gold[f"d{k}"], rtol=1e-5, atol=1e-5), k
    with pytest.raises(NotImplementedError):
        cfg = gen.mapper_cfg()
        cfg["mapping"]["BA"] = True
        Mapper(cfg, types.SimpleNamespace(device="cpu"), types.SimpleNamespace(
            bound=None, video=video, mapping_net=mapper.mapping_net, renderer=None, reload_map=torch.zeros(1).int(),
            H=32, W=48, fx=1, fy=1, cx=1, cy=1))


def test_ate_rmse_recovers_known_similarity():
    """ATE with Sim(3) alignment (src/slam.py:343-360 via evo; here Umeyama's closed form): a trajectory that differs
    from the reference by a similarity transform has zero error and the transform is recovered; noise gives the
    noise level; without scale correction a scaled trajectory keeps an error."""
    from go_slam_amd.eval_ate import ate_rmse, umeyama_alignment
    from go_slam_amd.lietorch_shim import SE3
    g = torch.Generator().manual_seed(2)
    ref = torch.cumsum(torch.randn(200, 3, generator=g, dtype=torch.float64) * 0.1, 0).numpy()
    Rm = SE3.exp(torch.tensor([[0.0, 0, 0, 0.4, -0.7, 0.2]], dtype=torch.float64)).matrix()[0, :3, :3].numpy()
    t, c = np.array([0.5, -1.0, 2.0]), 1.7
    est = ((ref - t) @ Rm) / c                                  # ref = c R est + t
    rmse, info = ate_rmse(est, ref)
    assert rmse < 1e-9 and abs(info["scale"] - c) < 1e-9
    assert np.allclose(info["rotation"], Rm, atol=1e-9) and np.allclose(info["translation"], t, atol=1e-9)
    R2, t2, c2 = umeyama_alignment(est.T, ref.T, with_scale=False)
    assert c2 == 1.0 and abs(np.linalg.det(R2) - 1.0) < 1e-9
    assert ate_rmse(est, ref, correct_scale=False)[0] > 0.05
    noisy = est + 0.01 * np.random.default_rng(0).standard_normal(est.shape)
    r = ate_rmse(noisy, ref)[0]
    assert 0.5 * 0.01 * c * 3 ** 0.5 < r < 1.5 * 0.01 * c * 3 ** 0.5
    mirrored = ref * np.array([1.0, 1.0, -1.0])                 # a reflection must not be "aligned away"
    assert ate_rmse(mirrored, ref)[0] > 0.05


def test_instant_neus_checkpoint_contract():
    """InstantNeuS keeps the reference module's state-dict keys, order and shapes (so `go.ckpt` / `mapping_net`
    checkpoints interchange, src/slam.py:296-301), the same split into network / volume parameter groups
    (src/mapping.py:52-58), and survives what src/slam.py / src/mesher.py do to it: share_memory(), copy.deepcopy,
    state_dict round trip (SURVEY 8b boundary 2)."""
    import copy
    from go_slam_amd.neus import InstantNeuS
    gold = np.load(os.path.join(HERE, "golden", "neus_forward.npz"))
    cfg = {"sdf_network": {"d_in": 3, "d_out": 32}, "color_network": {"d_in": 3, "d_feat": 31, "d_hidden": 64, "n_layers": 2},
           "variance_network": {"init_val": 0.2, "scale_factor": 10.0}, "sdf_smooth_std": 0.005,
           "sdf_sparse_factor": 5, "sdf_truncation": 0.16, "sdf_random_weight": 0.04}
    net = InstantNeuS(cfg, [[-2.5, 2.5]] * 3, device="cpu")
    sd = net.state_dict()
    assert list(sd.keys()) == [str(k) for k in gold["state_keys"]]
    assert [str(tuple(v.shape)) for v in sd.values()] == [str(s) for s in gold["state_shapes"]]
    names = {id(p): k for k, p in net.named_parameters()}
    assert [names[id(p)] for p in net.get_training_parameters()] == [str(k) for k in gold["train_param_names"]]
    assert [names[id(p)] for p in net.get_volume_parameters()] == [str(k) for k in gold["volume_param_names"]]
    net.share_memory()
    twin = copy.deepcopy(net)
    with torch.no_grad():
        twin.sdf_network.sdf_layer.weight.add_(1.0)
        twin.update_bound(torch.tensor([[-1.0, 1.0]] * 3))
    assert not torch.equal(twin.sdf_network.sdf_layer.weight, net.sdf_network.sdf_layer.weight)
    net.load_state_dict(twin.state_dict())
    assert torch.equal(net.sdf_network.sdf_layer.weight, twin.sdf_network.sdf_layer.weight)
    assert torch.equal(net.realtime_bound, torch.tensor([[-1.0, 1.0]] * 3))
    assert (net.sdf_truncation, net.sdf_sparse_factor) == (0.16, 5)


def test_wave_reduce_scatter_mapping_by_emulation():
    """common.h gs_wave_reduce_scatter (the BA kernels' block sums): emulate the six lane-bit steps on a 64-lane wave --
    a lane keeps the even or the odd element of each pair according to its lane bit and adds what its partner sends --
    and check the documented result: lane L ends with the wave total of value index 64 j + bitrev6(L) in v[j], for the
    two sizes the kernels use (90 and 42) and an odd one."""
    import numpy as np
    rng = np.random.default_rng(5)

    def bitrev6(x):
        return int(f"{x:06b}"[::-1], 2)

    for N in (90, 42, 7):
        vals = rng.standard_normal((64, N))                 # vals[lane][index]
        v = [list(vals[l]) for l in range(64)]
        n = N
        for mask in (32, 16, 8, 4, 2, 1):
            nn = (n + 1) // 2
            new = [[0.0] * nn for _ in range(64)]
            for lane in range(64):
                up = (lane & mask) != 0
                partner = lane ^ mask
                pup = (partner & mask) != 0
                for i in range(nn):
                    a = v[lane][2 * i]
                    b = v[lane][2 * i + 1] if 2 * i + 1 < n else 0.0
                    pa = v[partner][2 * i]
                    pb = v[partner][2 * i + 1] if 2 * i + 1 < n else 0.0
                    keep = b if up else a
                    recv = pa if pup else pb                 # the partner sends what IT does not keep
                    new[lane][i] = keep + recv
            v, n = new, nn
        total = vals.sum(0)
        for lane in range(64):
            for j in range(n):
                idx = 64 * j + bitrev6(lane)
                if idx < N:
                    assert abs(v[lane][j] - total[idx]) < 1e-9, (N, lane, j)
        assert n == (N + 63) // 64


def test_conv1x1_stage_row_arithmetic_is_exact():
    """conv1x1.hip stages a wave's 32 dense pixel rows through LDS and finds the row of an 8-byte unit u with
    (int)((u + 0.5f) * (1.0f / U)) instead of an integer division (U = K / 4 units per row): that must equal u // U for
    every supported K and every unit of the 32-row chunk; and the stage's row stride (an odd number of 16-byte pieces)
    must hold a row."""
    import numpy as np
    for K in range(4, 212, 4):
        U = K // 4
        u = np.arange(32 * U, dtype=np.int64)
        inv = np.float32(1.0) / np.float32(U)
        r = ((u.astype(np.float32) + np.float32(0.5)) * inv).astype(np.int64)
        assert np.array_equal(r, u // U), K
        pieces = (K + 7) // 8
        stride_halves = 8 * (pieces | 1)
        assert stride_halves >= K and (stride_halves // 8) % 2 == 1


def test_update_operator_context_term_cache_keys_and_eviction():
    """UpdateModule._edge_state keeps the hoisted context term per `inp` MEMORY (address, shape, strides, dtype) and
    validates it with the tensor's version counter: a fresh view of the same features hits (MotionFilter.track passes
    `self.inp[None]`, a new object per frame), an in-place write misses, different tensors get their own entries
    (update_lowmem's chunks), an entry lives exactly as long as the tensor that owns the features' memory (weak reference:
    the frontend replaces `graph.inp` on every edge-set change and must not pin the old ones; a new tensor that receives
    a freed address computes its own term), the LRU honours its entry / byte caps, and the debug checksum mode catches a
    write the version counter does not see.  The convolution itself is replaced by a counter here (no GPU)."""
    import gc
    from go_slam_amd.droid_net import UpdateModule
    op = UpdateModule()
    calls = []

    def fake_gates(x):
        calls.append(x.data_ptr())
        return x.float().sum().reshape(1)
    op.gru.inp_gates = fake_gates
    op.gru._half_weights = lambda: None
    op.gru._hw_key = 0
    n, h, w = 3, 4, 5
    base = torch.randn(n, 128, h, w).half().contiguous(memory_format=torch.channels_last)[None]
    _, t1 = op._edge_state(base, n, h, w)
    _, t2 = op._edge_state(base[0][None], n, h, w)              # a NEW view object of the same memory: hit
    assert len(calls) == 1 and t2 is t1
    base.mul_(2.0)                                              # written to: the version counter moves -> recomputed
    _, t3 = op._edge_state(base[0][None], n, h, w)
    assert len(calls) == 2 and float(t3) != float(t1)
    others = [torch.randn(n, 128, h, w).half().contiguous(memory_format=torch.channels_last)[None] for _ in range(5)]
    for o in others:
        op._edge_state(o, n, h, w)
    assert len(calls) == 7 and len(op._inp_pre_cache) == 6     # one entry per tensor, the first one still there
    for o in others:                                            # walking the chunks again: all hits
        op._edge_state(o, n, h, w)
    assert len(calls) == 7
    # an entry dies with the tensor that owns its memory: nothing is pinned, and the freed address serves nothing stale
    ptr, term = others[0].data_ptr(), op._inp_pre_cache[next(k for k in op._inp_pre_cache if k[0] == others[0].data_ptr())][3]
    del others[0], o
    gc.collect()
    assert len(op._inp_pre_cache) == 5 and all(k[0] != ptr for k in op._inp_pre_cache)
    fresh = [torch.randn(n, 128, h, w).half().contiguous(memory_format=torch.channels_last)[None] for _ in range(8)]
    for f in fresh:
        if f.data_ptr() == ptr:                                 # the allocator handed the address out again
            k = len(calls)
            _, tf = op._edge_state(f, n, h, w)
            assert len(calls) == k + 1 and tf is not term
    # debug checksum mode: a write through an alias with its own version counter (`.data`: what a raw-pointer write from a
    # HIP launch or another process looks like to torch) is served stale silently by default -- and raises in debug mode
    chk = UpdateModule()
    chk.gru.inp_gates, chk.gru._half_weights, chk.gru._hw_key = fake_gates, (lambda: None), 0
    victim = torch.randn(n, 128, h, w).half().contiguous(memory_format=torch.channels_last)[None]
    _, v1 = chk._edge_state(victim, n, h, w)
    v = victim._version
    victim.data.add_(1.0)
    assert victim._version == v
    _, v2 = chk._edge_state(victim, n, h, w)
    assert v2 is v1                                             # (the default trusts the version counter)
    chk.invalidate_context()                                    # the explicit hook for such writers
    _, v3 = chk._edge_state(victim, n, h, w)
    assert v3 is not v1 and float(v3) != float(v1)
    chk.cache_check = True
    chk.invalidate_context()
    chk._edge_state(victim, n, h, w)
    chk._edge_state(victim, n, h, w)                            # unchanged: fine
    victim.data.add_(1.0)
    with pytest.raises(RuntimeError, match="behind the version counter"):
        chk._edge_state(victim, n, h, w)
    # caps: at most INP_CACHE_ENTRIES entries, least recently used first; a weight change drops every old term
    op.INP_CACHE_ENTRIES = 3
    for f in fresh:
        op._edge_state(f, n, h, w)
    assert len(op._inp_pre_cache) == 3
    assert [k[0] for k in op._inp_pre_cache] == [f.data_ptr() for f in fresh[-3:]]
    op.gru._hw_key = 1
    op._edge_state(fresh[-1], n, h, w)
    assert len(op._inp_pre_cache) == 1
    op.drop_edge_caches()
    assert op._inp_pre_cache is None


def test_ray_bank_draws_what_build_rays_draws(monkeypatch):
    """neus/rays.RayBank (the frames of one Mapper call stacked once; a draw = the reference's randint calls + one
    searchsorted over the masks' running sums) against build_rays called frame by frame (src/nerf_func.py:115-181,
    src/mapping.py:222-240) under the same seed: same pixels -- origins, colours, depths bit for bit, directions to
    fp32 rounding of the 3 x 3 product -- with ragged masks, a frame without a mask, repeated frames, numpy poses; and
    the reference's own form when a mask leaves fewer than 2 n_rays pixels (then EVERY valid pixel is returned)."""
    from go_slam_amd.neus import rays as R
    g = torch.Generator().manual_seed(5)
    H, W, F = 24, 40, 5
    fx, fy, cx, cy = 30.0, 31.0, 19.5, 11.5
    items = {}
    for f in range(F):
        c2w = torch.eye(4)
        q = torch.linalg.qr(torch.randn(3, 3, generator=g))[0]
        c2w[:3, :3], c2w[:3, 3] = q, torch.randn(3, generator=g)
        mask = (torch.rand(H, W, generator=g) < (0.3 + 0.15 * f)).float() if f != 2 else None
        items[10 + f] = (torch.rand(H, W, 3, generator=g), torch.rand(H, W, generator=g) * 3 + 0.5,
                         c2w.numpy() if f == 1 else c2w, None, mask)
    bank = R.RayBank(items, H, W, fx, fy, cx, cy, "cpu")
    assert bank.N[2] == H * W and all(0 < n <= H * W for n in bank.N)
    frames = [12, 10, 14, 10, 13]

    def per_frame(n_rays):
        parts = [[], [], [], []]
        for f in frames:
            color, depth, c2w, _, mask = items[f]
            out = R.build_rays(0, H, 0, W, n_rays, H, W, fx, fy, cx, cy, c2w, depth, color, "cpu",
                               nerf_coordinate=False, dir_normalize=False, mask=mask)
            for acc, x in zip(parts, out):
                acc.append(x.float())
        o, d, dep, col = (torch.cat(p, 0) for p in parts)
        return o, d, col, dep
    torch.manual_seed(77)
    want = per_frame(37)
    real = R.build_rays
    monkeypatch.setattr(R, "build_rays", lambda *a, **k: (_ for _ in ()).throw(AssertionError("batched path expected")))
    torch.manual_seed(77)
    got = bank.sample(frames, 37)
    monkeypatch.setattr(R, "build_rays", real)
    assert got[0].shape == (37 * len(frames), 3)
    assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[2]) and torch.equal(got[3], want[3])
    torch.testing.assert_close(got[1], want[1], rtol=1e-6, atol=1e-6)
    torch.manual_seed(77)                                          # both consume the generator identically: the next
    per_frame(37)                                                  # draw after either is the same
    nxt = torch.rand(3)
    torch.manual_seed(77)
    bank.sample(frames, 37)
    assert torch.equal(torch.rand(3), nxt)
    # a mask with fewer than 2 n_rays pixels: the reference returns every valid pixel of that frame
    n_big = min(bank.N) // 2 + 1
    torch.manual_seed(78)
    want = per_frame(n_big)
    torch.manual_seed(78)
    got = bank.sample(frames, n_big)
    assert all(torch.equal(x, y) for x, y in zip(got, want))


def test_get_mapping_items_equals_the_per_keyframe_hand_outs():
    """DepthVideo.get_mapping_items (one pass for all keyframes a Mapper call visits) vs get_mapping_item called once per
    list entry (src/depth_video.py:153-177): the same five tensors per keyframe bit for bit, and the priorities decayed
    once per OCCURRENCE (the mapper's visit list repeats keyframes)."""
    gen = _gen()
    vids = []
    for _ in range(2):
        video = DepthVideo.from_config(gen.mapper_cfg(), types.SimpleNamespace(device="cpu"))
        gen.fill_mapping_video(video)
        video.pose_compensate[0] = torch.tensor([0.1, -0.2, 0.05, 0.0, 0.1, 0.0, 0.995]) / torch.tensor(
            [1, 1, 1, 1, 1, 1, 1.0])
        vids.append(video)
    order = [7, 3, 7, 5, 3, 7]
    one = {i: vids[0].get_mapping_item(i, "cpu", decay=0.5) for i in order}      # (a dict: later hand-outs overwrite)
    many = vids[1].get_mapping_items(order, "cpu", decay=0.5)
    assert list(many) == [7, 3, 5]
    for i in many:
        for a, b in zip(one[i], many[i]):
            assert torch.equal(a, b)
    assert torch.equal(vids[0].update_priority, vids[1].update_priority)
    assert vids[1].get_mapping_items([], "cpu") == {}


def test_depth_video_write_hooks_fire_on_feature_writes_and_stay_out_of_pickles():
    """DepthVideo.add_write_hook: the explicit invalidation point for caches built from `fmaps / nets / inps` by readers
    that cannot see torch's version counters move (another process on the shared buffers, a raw-pointer writer)."""
    import pickle
    from go_slam_amd.depth_video import DepthVideo
    v = DepthVideo(4, 6, buffer=5, device="cpu")
    seen = []

    class Reader:
        def forget(self, index):
            seen.append(("m", index))
    r = Reader()
    v.add_write_hook(r.forget)
    v.add_write_hook(lambda ix: seen.append(("f", ix)))
    img = torch.zeros(3, 32, 48)
    f = torch.zeros(1, 128, 4, 6).half()
    v[0] = (0.0, img, None, None, None, None)                              # no feature write: no hook
    assert seen == []
    v[1] = (1.0, img, None, None, None, None, f, f[0], f[0])
    assert seen == [("m", 1), ("f", 1)]
    del r                                                                  # bound methods are held weakly
    v[2] = (2.0, img, None, None, None, None, f, f[0], f[0])
    assert seen[2:] == [("f", 2)]
    w = pickle.loads(pickle.dumps({k: x for k, x in v.__getstate__().items() if not k.startswith("_counter") and k not in
                                   ("ready", "mapping", "ba_lock", "global_ba_lock")}))
    assert "_write_hooks" not in w
