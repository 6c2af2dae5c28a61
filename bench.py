#!/usr/bin/env python
"""Hot-path benchmark (contract: python bench.py --gpus N --steps K --warmup W -> ONE JSON line).

Metric (BASELINE.json): BA keyframes/s on synthetic 640x480 RGB-D.  One *step* = the
frontend's per-keyframe work unit = 6 x FactorGraph.update(iters=2, use_inactive=True)
(reference src/frontend.py:66-67,90-91) on a window graph of P=25 keyframes / E=75 edges at
the 1/8-resolution map size 60x80: reproject + 4-level corr lookup + UpdateModule (own implicit-GEMM
3x3 convolutions with fused ConvGRU epilogues) + 2 Gauss-Newton dense-BA iterations + convex upsampling.  Inputs are resident in HBM before the
timed region.  Tracking does not shard ("replicas only"), so --gpus N runs N
independent replicas, one process per GPU, and `value` is their aggregate; the mapping legs
(`neus_train`, `neus_train_weak`) shard their ray batch over the N ranks with RCCL collectives.

Launching: under an external launcher (`python -m torch.distributed.run ... bench.py --gpus N`, WORLD_SIZE / RANK /
LOCAL_RANK in the environment) every process is one rank.  Called plainly as `python bench.py --gpus N` with N > 1 and
no WORLD_SIZE, the script re-executes itself under `torch.distributed.run --nnodes=1 --nproc-per-node N
--master-addr 127.0.0.1` and passes rank 0's JSON line through; `n_gpus` is the RCCL world size either way and
`rccl_ranks` is the sum of ones the ranks all-reduced on their devices.

Extra objects on the line: `roofline` for the time-dominant hand-written kernel (the update
operator's implicit-GEMM 3x3 convolution, MFMA-bound: algorithmic FLOPs / launch time measured with
events on the launch stream; the HBM-bound lookup / volume kernels are in `roofline_other`) and `cpu_baseline` (the CPU oracle + the same UpdateModule on the host cores,
on a bounded 1-update sample, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F16_PEAK_TFLOPS = 2500.0   # dense fp16/bf16 MFMA peak (no sparsity)
SHAPE = "S480"
NUM_KF = 25
NUM_EDGES = 75
UPDATES_PER_KF = 6


def bench_graph(num_kf, num_edges, seed=43):
    """Frontend-like window graph: |i-j|==1 both directions, then random proximity pairs |i-j|<=6."""
    g = torch.Generator().manual_seed(seed)
    pairs = [(i, i + 1) for i in range(num_kf - 1)] + [(i + 1, i) for i in range(num_kf - 1)]
    have = set(pairs)
    while len(pairs) < num_edges:
        i = int(torch.randint(0, num_kf, (1,), generator=g))
        d = int(torch.randint(2, 7, (1,), generator=g)) * (1 if int(torch.randint(0, 2, (1,), generator=g)) else -1)
        j = i + d
        if 0 <= j < num_kf and (i, j) not in have:
            have.add((i, j))
            pairs.append((i, j))
    return (torch.tensor([p[0] for p in pairs], dtype=torch.long),
            torch.tensor([p[1] for p in pairs], dtype=torch.long))


def build_state(device, seed=43, num_kf=None, num_edges=None, shape=None, corr_impl="volume", upsample=True, rgbd=True):
    from go_slam_amd import synth
    from go_slam_amd.depth_video import DepthVideo
    from go_slam_amd.droid_net import UpdateModule
    from go_slam_amd.factor_graph import FactorGraph

    NUM_KF, NUM_EDGES, SHAPE = (num_kf or globals()["NUM_KF"], num_edges or globals()["NUM_EDGES"],
                                shape or globals()["SHAPE"])
    ht, wd, _ = synth.SHAPES[SHAPE]
    torch.manual_seed(seed)
    vid = synth.make_video(NUM_KF, SHAPE, seed=seed, rgbd=rgbd, buffer=NUM_KF + 7)
    video = DepthVideo(ht, wd, buffer=NUM_KF + 7, device=device)
    video.poses.copy_(vid["poses"])
    video.disps.copy_(vid["disps"])
    video.disps_sens.copy_(vid["disps_sens"])
    video.intrinsics.copy_(vid["intrinsics"])
    video.counter = NUM_KF
    g = torch.Generator().manual_seed(seed + 5)
    video.fmaps[:NUM_KF, 0] = torch.randn(NUM_KF, 128, ht, wd, generator=g).half().to(device)
    video.nets[:NUM_KF] = torch.tanh(torch.randn(NUM_KF, 128, ht, wd, generator=g)).half().to(device)
    video.inps[:NUM_KF] = torch.relu(torch.randn(NUM_KF, 128, ht, wd, generator=g)).half().to(device)
    update_op = UpdateModule().to(device).eval().to(memory_format=torch.channels_last)
    # small output heads so that the synthetic GRU state stays in a sane flow range
    with torch.no_grad():
        update_op.delta[2].weight.mul_(0.05)
        update_op.delta[2].bias.zero_()
    graph = FactorGraph(video, update_op, device=device, corr_impl=corr_impl, max_factors=NUM_EDGES, upsample=upsample)
    ii, jj = bench_graph(NUM_KF, NUM_EDGES, seed)
    graph.add_factors(ii.to(device), jj.to(device))
    return video, update_op, graph, (vid, ii, jj)


def keyframe_step(graph):
    # A real frontend changes the edge set at every keyframe, which invalidates what the package
    # caches per edge set (host-side edge indices, the hoisted context-feature convolutions of the
    # GRU).  The bench graph is static, so drop those caches here: every timed keyframe pays for
    # rebuilding them once, as a live run would.
    graph._eidx = None
    graph.update_op.drop_edge_caches()
    for _ in range(UPDATES_PER_KF):
        graph.update(None, None, use_inactive=True)


def global_ba_stress(device, num_kf=200, num_edges=1200, shape="Scan"):
    """BASELINE configs[3] stress shape: >= 200 keyframes, 1200 edges at 30x40 maps -- wall time of
    ONE full-BA `update_lowmem` step (alt-corr lookups in chunks of 13 source keyframes + update operator + a
    dense BA over all edges: 6P = 1194 unknowns, blocked multi-kernel Cholesky)."""
    video, update_op, graph, _ = build_state(device, seed=47, num_kf=num_kf, num_edges=num_edges, shape=shape,
                                             corr_impl="alt", upsample=False)
    # One backend invocation = `update_lowmem(steps=8)` (reference src/backend.py: 6-8 steps per global / loop BA) on a
    # graph whose per-edge caches are cold, as after the frontend added keyframes: the first step pays the hoisted
    # context-term convolutions of every chunk, the other seven reuse them.  Reported: the invocation's time / 8 (what
    # rounds 3-5 called the step: until this round every step paid everything), the cold first step and a steady one.
    STEPS = 8

    def invocation():
        graph.update_op.drop_edge_caches()
        graph.update_lowmem(steps=STEPS, iters=2)

    def first_step():
        graph.update_op.drop_edge_caches()
        graph.update_lowmem(steps=1, iters=2)
    ms = time_op(invocation, iters=2, warm=1) / STEPS
    ms_first = time_op(first_step, iters=2, warm=1)
    ms_steady = time_op(lambda: graph.update_lowmem(steps=1, iters=2), iters=3, warm=1)
    finite = bool(torch.isfinite(video.poses).all()) and bool(torch.isfinite(video.disps).all())
    out = {"keyframes": num_kf, "edges": int(graph.ii.numel()), "maps": shape, "unknowns": 6 * (num_kf - 1),
           "update_lowmem_step_ms": ms, "steps_per_invocation": STEPS, "first_step_ms": ms_first,
           "steady_step_ms": ms_steady, "state_finite": finite}
    # the on-the-fly correlation of the step, timed per launch by the library's kernel timer (HIP events on the launch
    # stream): SURVEY 8d prices it at 65,536 * HW flop per edge (4 levels x 64 taps x 128 channels x 2)
    from go_slam_amd import _lib
    with _lib.kernel_timer(device) as kt:
        graph.update_lowmem(steps=1, iters=2)
        torch.cuda.synchronize()
    t = kt.read()
    out["library_launches_per_step"] = int(sum(v[1] for v in t.values()))
    if "altcorr_pyramid" in t and t["altcorr_pyramid"][1]:
        tot_ms, n = t["altcorr_pyramid"]
        flops = 65536.0 * graph.ht * graph.wd * int(graph.ii.numel())
        tf = flops / (tot_ms * 1e-3) / 1e12
        out["altcorr_roofline"] = {
            "kernel": "altcorr_pyramid_kernel (4 levels x 4x4-pixel tiles x union window on v_mfma_f32_16x16x32_f16; "
                      f"{n} launches = one per 13-keyframe chunk) @ {int(graph.ii.numel())} edges, {graph.ht}x{graph.wd} maps",
            "bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / MFMA_F16_PEAK_TFLOPS,
            "algorithmic_flops_per_step": flops, "kernel_ms_per_step": tot_ms, "kernel_avg_us": 1e3 * tot_ms / n,
            "frac_of_fp32_vector_roof_157TF": tf / 157.0,
            "note": "algorithmic flops = the 64 taps each pixel needs; the dense union-window product issues ~3-4x as many "
                    "on the matrix cores (round 4: one wave per pixel on v_dot2, 16.4 TFLOP/s = 0.10 of the vector roof)"}
    return out


def mono_window(device, num_kf=50, num_edges=100, shape="Rep", steps=5, warm=2):
    """BASELINE configs[4]'s TRACKING side: the monocular frontend window of replica_mono.yaml:29,32 (window 50,
    max_factors 100) at the Replica map size 40x80, no depth prior (disps_sens = 0) -- one step = 6 x
    FactorGraph.update(iters=2) as in the headline, but on a window whose reduced camera system has 6P = 294
    unknowns (past the 192 the RGB-D window's LDS-resident Cholesky holds)."""
    from go_slam_amd import droid_backends as db
    video, update_op, graph, _ = build_state(device, seed=53, num_kf=num_kf, num_edges=num_edges, shape=shape, rgbd=False)
    assert not bool(video.disps_sens.any())
    for _ in range(warm):
        keyframe_step(graph)
    torch.cuda.synchronize()
    quiesce_gc()
    tic = time.perf_counter()
    for _ in range(steps):
        keyframe_step(graph)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - tic) / steps
    finite = bool(torch.isfinite(video.poses).all()) and bool(torch.isfinite(video.disps).all())
    ii, jj = graph.ii, graph.jj
    ht, wd = graph.ht, graph.wd
    target = graph.target.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
    weight = torch.rand_like(target)
    kx = torch.unique(torch.cat([torch.arange(1, num_kf, device=ii.device), ii]))
    eta = 0.2 * graph.damping[kx].contiguous() + 1e-7
    poses0, disps0 = video.poses.clone(), video.disps.clone()

    def ba():
        video.poses.copy_(poses0)
        video.disps.copy_(disps0)
        db.ba(video.poses, video.disps, video.intrinsics[0].contiguous(), video.disps_sens, target, weight, eta,
              ii, jj, 1, num_kf, 2, 1e-4, 0.1, False)
    ba_ms = time_op(ba)
    return {"workload": f"monocular frontend window: {shape} maps {ht}x{wd}, P={num_kf} keyframes, E={int(ii.numel())} edges, "
                        "no depth prior, 6 updates/keyframe, iters=2", "keyframes_per_s": 1e3 / ms, "ms_per_keyframe": ms,
            "ba_2iter_ms": ba_ms, "unknowns": 6 * (num_kf - 1), "state_finite": finite}


def quiesce_gc():
    """What main() does once before the headline's timed region, repeated before every other leg's: the objects built so
    far leave the cyclic collector's reach, so that a full collection (~20 ms on this process's heap) cannot land inside a
    timed loop of a few milliseconds -- one did, in a 10-step loop of the 4096-ray mapper leg: 2.76 ms per step instead
    of 0.76.  Garbage created from here on is still collected."""
    import gc
    gc.collect()
    gc.freeze()


def time_op(fn, iters=10, warm=2):
    for _ in range(warm):
        fn()
    quiesce_gc()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters   # ms


def op_breakdown(video, update_op, graph):
    """Per-op milliseconds (events on the launch stream) for one update's components."""
    from go_slam_amd import droid_backends as db
    ii, jj = graph.ii, graph.jj
    E = ii.numel()
    ht, wd = graph.ht, graph.wd
    coords1, _ = video.reproject(ii, jj)
    corr = graph.corr(coords1)
    motion = torch.cat([coords1 - graph.coords0, graph.target - coords1], dim=-1).permute(0, 1, 4, 2, 3).clamp(-64, 64)
    out = {}
    out["reproject_ms"] = time_op(lambda: video.reproject(ii, jj))
    out["corr_lookup_ms"] = time_op(lambda: graph.corr(coords1))
    # the production path: the lookup fused with corr_encoder[0] (196 -> 128, bias + ReLU), one launch
    if hasattr(graph.corr, "lookup_encoded") and graph.corr.fused_encoder_supported():
        wpad, bias = update_op._corr_enc0_padded()
        out["corr_lookup_enc0_ms"] = time_op(lambda: graph.corr.lookup_encoded(coords1, wpad, bias))

    def gru():
        with torch.no_grad(), torch.autocast("cuda", dtype=torch.float16):
            update_op(graph.net, graph.inp, corr, motion, ii, jj)
    out["update_module_ms"] = time_op(gru, iters=5)
    t0, t1 = 1, NUM_KF
    target = graph.target.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
    weight = torch.rand_like(target)
    kx = torch.unique(torch.cat([torch.arange(t0, t1, device=ii.device), ii]))
    eta = 0.2 * graph.damping[kx].contiguous() + 1e-7
    poses0, disps0 = video.poses.clone(), video.disps.clone()

    def ba():
        video.poses.copy_(poses0)
        video.disps.copy_(disps0)
        db.ba(video.poses, video.disps, video.intrinsics[0].contiguous(), video.disps_sens, target, weight, eta,
              ii, jj, t0, t1, 2, 1e-4, 0.1, False)
    out["ba_2iter_ms"] = time_op(ba)
    video.poses.copy_(poses0)
    video.disps.copy_(disps0)
    out["edges"] = E
    out["motion_filter_frame_ms"] = motion_filter_frame_ms(ii.device)
    # correlation-volume build (HBM-write bound): 8 new edges, as one add_factors call of a frontend keyframe
    nb = 8
    f1 = torch.randn(nb, 128, ht, wd, device=ii.device).half()
    f2 = torch.randn(nb, 128, ht, wd, device=ii.device).half()
    layout = db.CORR_TILE8 if db.corr_tile8_supported(f1) else db.CORR_ROWMAJOR
    out["corr_build_8edges_ms"] = time_op(lambda: db.corr_volume_pyramid(f1, f2, layout), iters=5)
    out["corr_build_bytes"] = nb * 2.0 * (ht * wd) * sum(
        int(db._lib.lib().gs_corr_level_elems(ht, wd, l, layout)) for l in range(4))
    return out


def encoder_tail_roofline(device):
    """gs_norm_act (csrc/instnorm.hip) on the feature encoder's layer-1 activation (1 x 240 x 320 x 32, NHWC fp16):
    conv bias + InstanceNorm + ReLU + residual add + ReLU in three launches.  HBM-bound: x read for the statistics, then
    x and skip read and y written = 4 x 2 B per element."""
    from go_slam_amd import extractor as EX
    n, c, h, w = 1, 32, 240, 320
    x = torch.randn(n, c, h, w, device=device).half().contiguous(memory_format=torch.channels_last)
    skip = torch.randn(n, c, h, w, device=device).half().contiguous(memory_format=torch.channels_last)
    bias = torch.randn(c, device=device).half()
    ms = time_op(lambda: EX._norm_act(x, skip, True, True, True, bias=bias), iters=20, warm=3)
    nbytes = 4 * 2.0 * n * c * h * w
    gbs = nbytes / (ms * 1e-3) / 1e9
    return {"kernel": "instnorm_stats + instnorm_final + norm_act (gs_norm_act: conv bias + InstanceNorm + ReLU + skip add + "
                      "ReLU of one encoder block at 240x320x32)", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes, "kernel_avg_us": ms * 1e3,
            "note": "three launches on a 4.9 MB tensor: bound by launch / dependency latency, not by HBM (DESIGN 3b)"}


def motion_filter_frame_ms(device, return_fn=False):
    """Per INPUT frame (not per keyframe): MotionFilter.track on a 480x640 RGB-D frame that is not promoted to a keyframe
    (src/motion_filter.py:41-90) = feature encoder fnet (7x7 stride-2 stem, residual blocks with instance norm; the
    reference's BasicEncoder, here gs_enc_conv + gs_norm_act, NHWC fp16) + correlation volume against the last keyframe +
    one update-operator iteration + the keyframe decision (one host scalar)."""
    from go_slam_amd.depth_video import DepthVideo
    from go_slam_amd.droid_net import DroidNet
    from go_slam_amd.motion_filter import MotionFilter
    torch.manual_seed(43)
    net = DroidNet().to(device).eval()
    video = DepthVideo(60, 80, buffer=8, device=device, full_res=True)
    mf = MotionFilter(net, video, thresh=1e9, device=device)        # never promotes after the first frame
    g = torch.Generator().manual_seed(44)
    img = torch.rand(1, 3, 480, 640, generator=g).to(device)
    depth = (torch.rand(480, 640, generator=g) * 3 + 1).to(device)
    intr = torch.tensor([577.59, 578.73, 318.91, 242.68], device=device)
    mf.track(0.0, img.clone(), depth, intr)                        # first frame: keyframe 0

    def frame():
        mf.track(1.0, img.clone(), depth, intr)
    if return_fn:
        return frame
    return time_op(frame, iters=10, warm=3)


def sequence_cfg(enable_loop=False, buffer=128, H=480, W=640):
    """configs/go_slam.yaml's tracking block (the values every Replica / ScanNet config inherits) at 480 x 640"""
    return {"mode": "rgbd", "verbose": False, "cam": {"H_out": H, "W_out": W},
            "tracking": {"buffer": buffer, "beta": 0.75, "warmup": 8, "upsample": True, "motion_filter": {"thresh": 4.0},
                         "frontend": {"enable_loop": enable_loop, "keyframe_thresh": 4.0, "thresh": 16.0, "window": 25,
                                      "radius": 1, "nms": 1, "max_factors": 75},
                         "backend": {"thresh": 25.0, "radius": 1, "nms": 5, "loop_window": 25, "loop_thresh": 25.0,
                                     "loop_radius": 1, "loop_nms": 12},
                         "multiview_filter": {"thresh": 0.01, "visible_num": 2, "kernel_size": 1, "bound_enlarge_scale": 1.1}}}


def synthetic_trajectory(n, step_m=0.012, step_deg=0.35, seed=43):
    """Keyframe poses on a smooth arc, stored world -> camera as [t, q] (SURVEY 8(d)'s trajectory -- a forward drift with a
    slow yaw and a gentle sideways weave).  The default step puts enough neighbours of a keyframe under the frontend's
    16 px proximity threshold at 2.5 m depth that the window runs AT its 75-edge cap (mean 64 active edges, max 75: the
    max_factors retirement of src/factor_graph.py:99-103 fires on most keyframes); step_m = 0.02 / step_deg = 0.6 gives the
    sparser regime (mean 46-49 edges, never at the cap)."""
    import math
    k = torch.arange(n, dtype=torch.float32)
    yaw = math.radians(step_deg) * k
    pos = torch.stack([step_m * k * 0.5 + 0.05 * torch.sin(0.2 * k), 0.02 * torch.sin(0.13 * k), step_m * k], 1)   # camera centres
    q = torch.stack([torch.zeros(n), torch.sin(yaw / 2), torch.zeros(n), torch.cos(yaw / 2)], 1)                   # camera -> world, about y
    # world -> camera: q_wc = conj(q), t_wc = -R(q_wc) pos
    qc = q * torch.tensor([-1.0, -1.0, -1.0, 1.0])
    v, w = qc[:, :3], qc[:, 3:]
    t = pos + 2.0 * torch.cross(v, torch.cross(v, pos, dim=1) + w * pos, dim=1)
    return torch.cat([-t, qc], 1)


def sequence_bench(device, keyframes=40, warm_keyframes=34, frames_per_keyframe=4, enable_loop=False, shared_video=False,
                   drop_every=8, return_state=False, spare_keyframes=4, freeze_gc=False, step_m=0.012, step_deg=0.35):
    """The tracker END TO END on a synthetic 640 x 480 RGB-D sequence, steady state: per input frame `MotionFilter.track`
    (src/motion_filter.py:39-90: feature encoder, correlation against the last keyframe, one update iteration, the
    keyframe decision), per keyframe `Frontend.__call__` (src/frontend.py:48-104: retire old edges, `add_proximity_factors`
    = frame_distance + NMS proposal, correlation volumes for the new edges, 4 updates, the keyframe-distance read,
    `rm_keyframe` or 2 more updates -- or, with `enable_loop` and more than `window` keyframes, `loop_ba`), with
    configs/go_slam.yaml's tracking parameters (window 25, max_factors 75, warm-up 8, upsample on).

    A random-weight network cannot track, so two decisions are SCRIPTED while all their work still runs: every
    `frames_per_keyframe`-th frame is promoted (`MotionFilter.thresh` is set to -1 / 1e9 around the call; the flow estimate
    and its host read happen every frame), and `Frontend._moved_enough` computes and reads the real keyframe distance but
    answers by script: every `drop_every`-th keyframe is dropped through `rm_keyframe`, the others are kept.  The pose of
    each new keyframe is seeded from the synthetic arc (instead of the previous keyframe's pose), so frame distances, the
    proximity proposals and the window's edge count are those of a moving camera.  Timed per keyframe with a device
    synchronisation between the frame stage and the keyframe stage (2 per keyframe)."""
    import types
    from go_slam_amd.depth_video import DepthVideo
    from go_slam_amd.droid_net import DroidNet
    from go_slam_amd.frontend import Frontend
    from go_slam_amd.motion_filter import MotionFilter
    torch.manual_seed(43)
    total = keyframes + warm_keyframes + spare_keyframes
    cfg, args = sequence_cfg(enable_loop, buffer=total + 8), types.SimpleNamespace(device=str(device))
    net = DroidNet().to(device).eval()
    with torch.no_grad():                               # small output heads: the synthetic state stays in a sane flow range
        net.update.delta[2].weight.mul_(0.05)
        net.update.delta[2].bias.zero_()
    video = DepthVideo.from_config(cfg, args) if shared_video else \
        DepthVideo(60, 80, buffer=total + 8, device=device, full_res=True)
    mf = MotionFilter(net, video, thresh=1e9, device=str(device))

    class ScriptedFrontend(Frontend):
        decisions = []

        def _moved_enough(self):
            super()._moved_enough()                     # the distance launch pair + the host read of the real frontend
            keep = (self.count % drop_every) != 0
            self.decisions.append(keep)
            return keep
    fe = ScriptedFrontend(net, video, args, cfg)
    gt = synthetic_trajectory(total + 8, step_m=step_m, step_deg=step_deg).to(device)
    g = torch.Generator().manual_seed(44)
    frames = [torch.rand(1, 3, 480, 640, generator=g).to(device) for _ in range(4)]
    vv, uu = torch.meshgrid(torch.arange(480.0), torch.arange(640.0), indexing="ij")
    depth = (2.5 + 1.0 * torch.sin(uu * 0.013) * torch.cos(vv * 0.017)).to(device)
    intr = torch.tensor([577.59, 578.73, 318.91, 242.68], device=device)
    stamp = [0]

    def frame_stage():
        for k in range(frames_per_keyframe):
            mf.thresh = -1.0 if k == frames_per_keyframe - 1 else 1e9
            n0 = video.counter.value
            mf.track(float(stamp[0]), frames[stamp[0] % 4].clone(), depth, intr)
            stamp[0] += 1
            if video.counter.value > n0:                # promoted: seed the new keyframe's pose from the arc
                video.poses[n0] = gt[min(stamp[0] // frames_per_keyframe, gt.shape[0] - 1)]
    t_frames = t_front = 0.0
    edges, kept = [], 0
    import gc
    for k in range(warm_keyframes + keyframes):
        if k == warm_keyframes:                         # (see quiesce_gc)
            quiesce_gc()
        timed = k >= warm_keyframes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        frame_stage()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        fe()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        if timed:
            t_frames += t1 - t0
            t_front += t2 - t1
            edges.append(int(fe.graph.ii.numel()))
            kept += int(bool(fe.decisions and fe.decisions[-1]))
    n = video.counter.value
    finite = bool(torch.isfinite(video.poses[:n]).all()) and bool(torch.isfinite(video.disps[:n]).all())
    # the headline's unit -- 6 x FactorGraph.update, per-edge-set caches dropped once (keyframe_step) -- on THIS graph as the
    # sequence left it (its active + inactive edges, its window): what the end-to-end keyframe is to be compared with
    g_ = fe.graph

    def unit():
        g_._eidx = None
        g_.update_op.drop_edge_caches()
        for _ in range(UPDATES_PER_KF):
            g_.update(None, None, use_inactive=True)
    p0, d0 = video.poses.clone(), video.disps.clone()
    unit_ms = time_op(unit, iters=5, warm=2)
    video.poses.copy_(p0)
    video.disps.copy_(d0)
    out = {"workload": f"{(warm_keyframes + keyframes) * frames_per_keyframe} frames of 480x640 RGB-D, a keyframe every "
                       f"{frames_per_keyframe} frames, {keyframes} timed keyframes after {warm_keyframes}; window 25, "
                       f"max_factors 75, enable_loop {enable_loop}, every {drop_every}th keyframe dropped (rm_keyframe)",
           "video": "shared (reference constructor: locked sections end with a stream sync)" if shared_video
                    else "single process (asynchronous)",
           "frontend_e2e_ms_per_keyframe": 1e3 * t_front / keyframes,
           "six_update_unit_ms_on_the_final_graph": unit_ms, "final_graph_edges": int(g_.ii.numel()),
           "e2e_over_unit": (1e3 * t_front / keyframes) / unit_ms,
           "note_on_the_ratio": "the unit is timed on the graph as the LAST keyframe left it (final_graph_edges), the end-to-end "
                                "figure averages over keyframes with edges_mean active edges: compare the edge counts before the ratio",
           "motion_filter_ms_per_frame": 1e3 * t_frames / (keyframes * frames_per_keyframe),
           "ms_per_keyframe_all": 1e3 * (t_front + t_frames) / keyframes,
           "frames_per_s": keyframes * frames_per_keyframe / (t_front + t_frames),
           "keyframes_per_s": keyframes / (t_front + t_frames),
           "edges_mean": sum(edges) / len(edges), "edges_max": max(edges), "inactive_edges": int(fe.graph.ii_inac.numel()),
           "keyframes_kept": kept, "keyframes_dropped": keyframes - kept, "keyframes_in_video": n, "state_finite": finite,
           "control_flow": "scripted keyframe decisions; every kernel and host read of the real loop runs"}
    if return_state:
        return out, (net, video, fe, mf, frame_stage)
    return out


def backend_bench(device, num_kf=200, steps=8):
    """`Backend.dense_ba(0, t, steps=8)` (src/backend.py:20-44,122-136) on a 200-keyframe ScanNet-shaped video (configs[3]):
    frame-distance matrix of all pairs, the edge proposal with NMS on the device, a fresh alt-correlation FactorGraph,
    `update_lowmem(steps)`.  Wall time of the whole call (a second call: the update operator's weights are packed)."""
    import types
    from go_slam_amd import synth
    from go_slam_amd.backend import Backend
    from go_slam_amd.depth_video import DepthVideo
    from go_slam_amd.droid_net import DroidNet
    ht, wd, _ = synth.SHAPES["Scan"]
    torch.manual_seed(47)
    vid = synth.make_video(num_kf, "Scan", seed=47, rgbd=True, buffer=num_kf + 7)
    video = DepthVideo(ht, wd, buffer=num_kf + 7, device=device)
    for k in ("poses", "disps", "disps_sens", "intrinsics"):
        getattr(video, k).copy_(vid[k])
    video.counter = num_kf
    g = torch.Generator().manual_seed(52)
    video.fmaps[:num_kf, 0] = torch.randn(num_kf, 128, ht, wd, generator=g).half().to(device)
    video.nets[:num_kf] = torch.randn(num_kf, 128, ht, wd, generator=g).tanh().half().to(device)
    video.inps[:num_kf] = torch.randn(num_kf, 128, ht, wd, generator=g).relu().half().to(device)
    net = DroidNet().to(device).eval()
    with torch.no_grad():
        net.update.delta[2].weight.mul_(0.05)
        net.update.delta[2].bias.zero_()
    cfg, args = sequence_cfg(False, buffer=num_kf + 7, H=8 * ht, W=8 * wd), types.SimpleNamespace(device=str(device))
    cfg["tracking"]["upsample"] = False
    be = Backend(net, video, args, cfg)
    p0, d0 = video.poses.clone(), video.disps.clone()
    res = {}

    def call():
        video.poses.copy_(p0)
        video.disps.copy_(d0)
        res["n"] = be.dense_ba(0, num_kf, steps=steps)
    ms = time_op(call, iters=2, warm=1)
    return {"keyframes": res["n"][0], "edges": int(res["n"][1]), "steps": steps, "global_ba_ms": ms,
            "ms_per_step": ms / steps, "maps": "Scan 30x40",
            "state_finite": bool(torch.isfinite(video.poses).all()) and bool(torch.isfinite(video.disps).all())}


def mapper_call_bench(device, use_bank=True, n_kf=20, iters=20, H=480, W=640):
    """A steady `Mapper.__call__` (src/mapping.py:151-302) on 20 filtered keyframes of 480 x 640, mapping.pixels 4400,
    window 16, `iters` joint iterations per call (configs/go_slam.yaml's mapping block): ms per joint iteration, hand-out
    and ray draws included."""
    import types
    import numpy as np
    import go_slam_amd.neus as N
    from go_slam_amd.depth_video import DepthVideo
    from go_slam_amd.neus import mapping as M
    dev = str(device)
    cfg = {"mode": "rgbd", "cam": {"H_out": H, "W_out": W}, "tracking": {"buffer": 32},
           "mapping": {"device": dev, "iters": iters, "decay": 0.5, "w_color_loss": 2.0, "w_sdf_loss": 2.0,
                       "w_eikonal_loss": 0.1, "uncertainty_weight_loss": True, "BA": False, "BA_cam_lr": 1e-3, "pixels": 4400,
                       "mapping_window_size": 16, "net_lr": 1e-3, "grid_lr": 1e-2}}
    args = types.SimpleNamespace(device=dev)
    torch.manual_seed(3)
    np.random.seed(3)
    video = DepthVideo.from_config(cfg, args)
    g = torch.Generator().manual_seed(7)
    v, u = torch.meshgrid(torch.arange(float(H)), torch.arange(float(W)), indexing="ij")
    depth = (2.0 + 0.2 * torch.sin(u * 0.02) * torch.cos(v * 0.03)).to(dev)
    video.images[:n_kf] = torch.rand(n_kf, 3, H, W, generator=g).to(dev)
    video.disps_filtered[:n_kf] = 1.0 / depth
    video.mask_filtered[:n_kf] = (torch.rand(n_kf, H, W, generator=g) < 0.9).float().to(dev)
    video.poses_filtered[:n_kf, 0] = 0.02 * torch.arange(n_kf, device=dev)
    video.update_priority[:n_kf] = 1.0
    video.timestamp[:n_kf] = torch.arange(n_kf, device=dev).float()
    video.bound[0] = torch.tensor([[-2.4, 2.4], [-2.4, 2.4], [-0.4, 2.4]], device=dev)
    video.filtered_id[0] = n_kf
    model = N.InstantNeuS({}, [[-2.5, 2.5]] * 3, device=dev).to(dev)
    slam = types.SimpleNamespace(verbose=False, bound=model.bound, video=video, mapping_net=model,
                                 renderer=N.Renderer(N_samples=24, N_surface=48), reload_map=torch.zeros(1).int(), H=H, W=W,
                                 fx=577.6, fy=578.7, cx=318.9, cy=242.7)
    mapper = M.Mapper(cfg, args, slam)
    mapper.use_ray_bank = use_bank
    for _ in range(4):                                       # the first call (10 x iterations on the new keyframes) and three
        mapper()                                             # steady ones: every ray-batch shape of the window has its graph
    torch.cuda.synchronize()
    g0 = mapper.global_step
    t = time.perf_counter()
    for _ in range(3):                                       # steady calls: `iters` joint iterations on the window each
        mapper()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
    its = max(1, mapper.global_step - g0)
    return {"mapper_ms_per_joint_iteration": 1e3 * dt / its, "iterations_timed": its, "rays_per_iteration": 4400,
            "keyframes": n_kf, "ray_draw": "RayBank" if use_bank else "build_rays per frame"}


def neus_render_bench(device, n_rays=4096, iters=20):
    """Second half of the BASELINE metric: NeuS render rays/s = N / wall time of
    Renderer.render_batch_ray (+ InstantNeuS.forward) for N rays x 72 samples, forward only."""
    import go_slam_amd.neus as neus
    g = torch.Generator().manual_seed(43)
    model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(device)
    with torch.no_grad():       # non-degenerate ("trained-like") field, SURVEY 8(d)
        model.sdf_network.encoding.encoding.params.copy_((torch.rand(model.sdf_network.encoding.encoding.params.shape, generator=g) - 0.5) * 0.1)
        model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(device) * 0.1
    R = neus.Renderer(N_samples=24, N_surface=48)
    o = (torch.rand(n_rays, 3, generator=g) * 6 - 3).to(device)
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1).to(device)
    gt = torch.rand(n_rays, generator=g) * 3.5 + 0.5
    gt[torch.rand(n_rays, generator=g) < 0.1] = 0
    gt = gt.to(device)
    with torch.no_grad():
        # `InstantNeuS.render_rays` is the entry point north_star names: Renderer.render_batch_ray with the network as the
        # receiver (tests/test_neus_gpu.py: bit-identical to the Renderer call, and vs the oracle at 4096 rays)
        ms = time_op(lambda: model.render_rays(o, d, gt, renderer=R), iters=iters, warm=3)
        z, dist = R.sample(o, d, model.bound, gt)
        ms_fwd = time_op(lambda: model(o, d, z, dist), iters=iters, warm=3)
    pts = n_rays * 72
    with torch.no_grad():       # the fused colour MLP alone (MFMA): 2*(80*64 + 64*64 + 64*16) flop per point, padded sizes
        x = torch.randn(pts, 67, device=device).half()
        ms_mlp = time_op(lambda: model.color_network.network(x), iters=iters, warm=3)
    mlp_tflops = pts * 2.0 * (80 * 64 + 64 * 64 + 64 * 16) / (ms_mlp * 1e-3) / 1e12
    return {"metric": "NeuS render rays/s (InstantNeuS.render_rays = Renderer.render_batch_ray + InstantNeuS.forward, 72 samples/ray)",
            "mlp_mfma": {"kernel": "neus_mlp_kernel (67->64->64->3 fused, fp16 MFMA)", "ms": ms_mlp,
                         "achieved": mlp_tflops, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": mlp_tflops / MFMA_F16_PEAK_TFLOPS,
                         "note": "includes the tcnn-compat wrapper's pad/launch; 295k points x 20.5 kflop is far too little work to fill the MFMA pipes"},
            "value": n_rays / (ms * 1e-3), "unit": "rays/s", "rays_per_batch": n_rays, "ms_per_batch": ms,
            "forward_ms": ms_fwd, "dtype": "f16 grid/MLP, f32 elsewhere",
            # 512 B of grid gathers per sample point (L2 / Infinity-Cache resident table)
            "gather_GBps": pts * 512.0 / (ms_fwd * 1e-3) / 1e9}


def neus_train_bench(device, rank, world, steps=10, warm=3, global_rays=32768, scaling="strong", sharded=None):
    """Mapping step (render + losses + backward + grad all-reduce + clip + AdamW) on a GLOBAL batch of
    `global_rays` rays x 72 samples sharded over `world` GPUs (BASELINE configs[4]).  Called twice: strong scaling
    (32768 rays in total) and weak scaling (4096 rays per GPU -- the reference mapper's own per-iteration batch,
    so 8 GPUs carry the 32768-ray batch of configs[4])."""
    import go_slam_amd.neus as neus
    from go_slam_amd.neus.mapper import MapTrainer
    g = torch.Generator().manual_seed(43)
    model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(device)
    with torch.no_grad():
        model.sdf_network.encoding.encoding.params.copy_((torch.rand(model.sdf_network.encoding.encoding.params.shape, generator=g) - 0.5) * 0.02)
        model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(device) * 0.1
    R = neus.Renderer(N_samples=24, N_surface=48)
    n = global_rays
    o = (torch.rand(n, 3, generator=g) * 6 - 3).to(device)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(device)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[torch.rand(n, generator=g) < 0.1] = 0
    gt = gt.to(device)
    col = torch.rand(n, 3, generator=g).to(device)
    pr = torch.rand(24, generator=g).to(device)
    tr = MapTrainer(model, R, rank=rank, world=world, sharded=sharded)
    import torch.distributed as dist
    sharded = bool(getattr(tr, "sharded", world > 1))

    def sync():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
    for _ in range(warm):
        loss = tr.step(o, d, col, gt, pr)
    sync()
    quiesce_gc()
    tic = time.perf_counter()
    for _ in range(steps):
        loss = tr.step(o, d, col, gt, pr)
    sync()
    dt = time.perf_counter() - tic
    exch = None
    if sharded:
        if world > 1:
            t = torch.tensor([dt], device=device, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
        if getattr(tr, "fused", False):
            # a second, instrumented pass (outside the timed region): events around the collectives on the step's stream ->
            # per rank the time the compute stream was BLOCKED on an exchange; compute = step - exposed exchange
            from go_slam_amd.neus.distributed import ExchangeTimer
            et = ExchangeTimer(device)
            tr.flat._exchange_timer = et
            sync()
            tic2 = time.perf_counter()
            for _ in range(steps):
                tr.step(o, d, col, gt, pr)
            tr.flat.wait_gather()
            sync()
            step2 = 1e3 * (time.perf_counter() - tic2) / steps
            tr.flat._exchange_timer = None
            ex = et.exposed_ms()
            mine = torch.tensor([step2, ex["total"] / steps, ex["reduce_scatter_and_dense_allreduce"] / steps,
                                 ex["clip_norm_allreduce"] / steps, ex["all_gather_wait"] / steps],
                                device=device, dtype=torch.float64)
            allr = torch.zeros(world, mine.numel(), device=device, dtype=torch.float64)   # (all_reduce of a one-hot-row
            allr[rank] = mine                                                             # matrix: works on RCCL and gloo)
            dist.all_reduce(allr)
            rows = [[round(float(v), 4) for v in r] for r in allr]
            exch = {"columns": ["step_ms", "exposed_exchange_ms", "reduce_scatter+dense_allreduce_ms", "clip_norm_allreduce_ms",
                                "all_gather_wait_ms"], "per_rank": rows,
                    "compute_ms_per_rank": [round(r[0] - r[1], 4) for r in rows],
                    "deferred_all_gather": bool(tr.flat.overlap_gather),
                    "note": "instrumented pass after the timed region (one event pair per collective phase on the step's "
                            "stream); exposed = time the compute stream was blocked on the exchange"}
    ms = 1e3 * dt / steps
    return {"metric": "NeuS mapping train step rays/s (render + loss + backward + all-reduce + clip + AdamW)",
            "value": n / (ms * 1e-3), "unit": "rays/s", "global_rays": n, "rays_per_gpu": n // world, "ms_per_step": ms,
            "scaling": scaling, "fused_step": bool(getattr(tr, "fused", False)),
            "graph_replay": bool(getattr(tr, "graph", False)),
            "collective_bytes_sent_per_rank": (tr.flat.collective_bytes() if tr.fused
                                               else (0 if world == 1 else 4 * sum(p.numel() for p in tr.train_params))),
            "exchange": "none" if not sharded else ("reduce-scatter(fp16 table grad) + sharded AdamW + all-gather(fp16 "
                                                    "table)" if tr.fused else "all-reduce(fp32 flat grad)"),
            "exchange_exposed_ms": (max(r[1] for r in exch["per_rank"]) if exch else (0.0 if not sharded else None)),
            "exchange_per_rank": exch, "final_loss": float(loss),
            "collectives_captured": any(bool(e.get("one")) for e in tr._graphs.values())
            if (sharded and getattr(tr, "fused", False)) else None,
            "capture_error": getattr(tr, "capture_collectives_error", None)}


def cpu_baseline(sample_updates=1):
    """The CPU oracle (+ the same UpdateModule on the host, fp32) on a bounded sample of the same
    workload: `sample_updates` update calls = sample_updates/6 keyframe."""
    from go_slam_amd import synth
    from go_slam_amd.droid_net import UpdateModule
    from oracle import droid_oracle as O

    torch.manual_seed(43)
    ht, wd, _ = synth.SHAPES[SHAPE]
    vid = synth.make_video(NUM_KF, SHAPE, seed=43, rgbd=True, buffer=NUM_KF + 7)
    ii, jj = bench_graph(NUM_KF, NUM_EDGES, 43)
    g = torch.Generator().manual_seed(48)
    fmaps = torch.randn(NUM_KF, 128, ht, wd, generator=g).half()
    net = torch.tanh(torch.randn(NUM_KF, 128, ht, wd, generator=g))[ii][None]
    inp = torch.relu(torch.randn(NUM_KF, 128, ht, wd, generator=g))[ii][None]
    op = UpdateModule().eval()
    coords0 = torch.stack(torch.meshgrid(torch.arange(wd).float(), torch.arange(ht).float(), indexing="xy"), -1)
    # the pyramid build is setup (not part of an update), done once outside the timed region
    pyr = O.corr_pyramid(fmaps[ii][None], fmaps[jj][None])
    poses, disps = vid["poses"].clone(), vid["disps"].clone()
    target, _ = O.reproject(poses, disps, vid["intrinsics"], ii, jj)
    K = vid["intrinsics"][0].contiguous()
    t0, t1 = 1, NUM_KF
    quiesce_gc()
    tic = time.perf_counter()
    with torch.no_grad():
        for _ in range(sample_updates):
            coords1, _ = O.reproject(poses, disps, vid["intrinsics"], ii, jj)
            motion = torch.cat([coords1 - coords0, target - coords1], -1).permute(0, 1, 4, 2, 3).clamp(-64, 64)
            corr = O.corr_lookup(pyr, coords1, 3).float()
            net, delta, weight, damping, upmask = op(net, inp, corr, motion, ii, jj)
            target = coords1 + delta
            kx = torch.unique(torch.cat([torch.arange(t0, t1), ii]))
            assert damping.shape[1] == len(kx)     # every keyframe of the window is an edge source
            eta = 0.2 * damping[0].contiguous() + 1e-7
            tg = target.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
            wg = weight.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
            O.ba(poses, disps, K, vid["disps_sens"], tg, wg, eta, ii, jj, t0, t1, 2, 1e-4, 0.1, False)
            disps.clamp_(min=0.001)
    dt = time.perf_counter() - tic
    kf_per_s = (sample_updates / UPDATES_PER_KF) / dt
    return {"value": kf_per_s, "unit": "keyframes/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{sample_updates} update(s) = {sample_updates}/{UPDATES_PER_KF} keyframe, {dt:.1f} s of CPU work "
                      f"(oracle reproject+lookup+BA, UpdateModule fp32 on CPU torch {torch.__version__})"}


def cpu_baseline_neus(n_rays=4096):
    """Path M on the host cores, same ray distribution and sizes as the `neus_render` / `neus_train_weak` legs: the CPU
    oracle's Renderer.render_batch_ray + InstantNeuS.forward (oracle/neus_oracle.py) for one 4096-ray x 72-sample
    batch, and ONE mapper iteration (forward + losses + torch.autograd backward + clip + AdamW on the differentiable
    restatement, oracle/neus_autograd.py) on the same batch.  kind "port": tiny-cuda-nn has no CPU path."""
    from oracle import neus_autograd as NA, neus_oracle as NO
    g = torch.Generator().manual_seed(43)
    P = NO.make_params(43, grid_init=0.05)
    o = torch.rand(n_rays, 3, generator=g) * 6 - 3
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1)
    gt = torch.rand(n_rays, generator=g) * 3.5 + 0.5
    gt[torch.rand(n_rays, generator=g) < 0.1] = 0
    col = torch.rand(n_rays, 3, generator=g)
    pr = torch.rand(24, generator=g)
    with torch.no_grad():
        quiesce_gc()
        tic = time.perf_counter()
        z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, pr)
        NO.neus_forward(o, d, z, dist, P)
        t_render = time.perf_counter() - tic
    names = ("grid", "sdf_w", "sdf_b", "color_B", "mlp")
    Pd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    Pd["variance"] = torch.tensor(float(P["variance"]), requires_grad=True)
    params = [Pd[k] for k in names] + [Pd["variance"]]
    opt = torch.optim.AdamW([{"params": params[1:], "lr": 1e-3}, {"params": params[:1], "lr": 1e-2}],
                            betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    quiesce_gc()
    tic = time.perf_counter()
    z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, pr)
    loss = NA.mapping_loss(NA.neus_forward_diff(o, d, z, dist, Pd), col, gt)
    loss.backward()
    torch.nn.utils.clip_grad_norm_(params, 35.0)
    opt.step()
    t_train = time.perf_counter() - tic
    cores = torch.get_num_threads()
    return {"neus_render": {"value": n_rays / t_render, "unit": "rays/s", "cores": cores, "kind": "port",
                            "sample": f"1 batch of {n_rays} rays x 72 samples, {t_render:.1f} s of CPU work (oracle "
                                      "render_sample + neus_forward, fp32 torch on the host)"},
            "neus_train": {"value": n_rays / t_train, "unit": "rays/s", "cores": cores, "kind": "port",
                           "sample": f"1 mapper iteration on {n_rays} rays x 72 samples, {t_train:.1f} s of CPU work "
                                     "(oracle forward + losses + torch.autograd backward + clip + AdamW)"}}


def gather_ceiling(n_rays):
    """Gather-only replay of the hash grid's OWN index stream on this batch (tools/gather_replay.py on an MI355X, committed:
    profiles/r05_gather_replay.json): what the table loads of a forward cost with no arithmetic around them -- level-major
    for the hashed levels + point-major for the dense ones, the order the round-5 forward issues them in.  (Round 4's
    comparator drew independent random points and the production kernel beat it by up to 1.8x: no ceiling.)"""
    path = os.path.join(ROOT, "profiles", "r05_gather_replay.json")
    if not os.path.exists(path):
        return None
    b = json.load(open(path)).get("batches", {}).get(str(n_rays))
    if not b:
        return None
    return dict(b, source="profiles/r05_gather_replay.json (tools/gather_replay.py, committed run)")


def neus_kernel_rooflines(device, n_rays, steps=5):
    """The kernels that own path M, timed LIVE with HIP events on the launch stream (the library's kernel timer,
    gs_timing_*): the eager fused mapper step on `n_rays` rays x 72 samples.  Algorithmic bytes per point:
    forward 512 B of grid gathers + 12 B in = 524 B -- since round 5 split over neus_encode_levels_kernel (the 11 hashed
    levels, level-major: 352 B of gathers + 8 B in) and neus_point_kernel (the 5 dense levels: 160 B, + the records);
    backward 512 B gathers + 512 B table-gradient scatter + 12 B = 1036 B (the production backward streams 256 B per point
    of forward records instead of gathering; the algorithmic count stays SURVEY's); the bin reduce reads the backward's
    record queues (6 B per record, 128 records per point at most) and writes the 11 hashed levels once.  `gather_rate` =
    table loads of the IN-BOUND points / kernel time, against the gather-only replay of the same index stream
    (`gather_frac_of_replay` = replay time / kernel time <= 1 by construction of the comparator).  Below the level-major
    crossover (gs_neus_level_major_min_points: the 4096-ray batch) the forward is the point kernel alone, gathering every
    level itself; its replay leg is then `point_major_16`."""
    import go_slam_amd.neus as neus
    from go_slam_amd import _lib
    from go_slam_amd.neus.mapper import MapTrainer
    g = torch.Generator().manual_seed(43)
    model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(device)
    with torch.no_grad():
        model.sdf_network.encoding.encoding.params.copy_((torch.rand(model.sdf_network.encoding.encoding.params.shape, generator=g) - 0.5) * 0.02)
        model.sdf_network.sdf_layer.weight[:, 3:] = torch.randn(32, 32, generator=g).to(device) * 0.1
    n = n_rays
    o = (torch.rand(n, 3, generator=g) * 6 - 3).to(device)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(device)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[torch.rand(n, generator=g) < 0.1] = 0
    gt = gt.to(device)
    col = torch.rand(n, 3, generator=g).to(device)
    pr = torch.rand(24, generator=g).to(device)
    tr = MapTrainer(model, neus.Renderer(N_samples=24, N_surface=48), graph=False)
    for _ in range(2):
        tr.step(o, d, col, gt, pr)
    torch.cuda.synchronize()
    with _lib.kernel_timer(device) as kt:
        for _ in range(steps):
            tr.step(o, d, col, gt, pr)
        torch.cuda.synchronize()
    t = kt.read()
    pts = n * 72
    ceil_ = gather_ceiling(n_rays)
    inb = ceil_["points_in_bound"] if ceil_ else pts
    pmc = {}
    ppath = os.path.join(ROOT, "profiles", "r06_pmc_neus.json")
    if os.path.exists(ppath):
        pmc = json.load(open(ppath)).get(f"traffic_bytes_per_launch@{n_rays}", {})
    nh = 11                         # hashed levels (2^19 entries x 4 B each)
    # (timer keys, label, algorithmic bytes, table loads per in-bound point, replay leg)
    spec = [(("neus_encode_levels", "neus_point"),
             "forward = neus_encode_levels_kernel (11 hashed levels, level-major, XCD-consecutive) + neus_point_kernel (dense "
             "levels + SDF linear + analytic gradient + alpha + colour MLP on MFMA)", 524.0 * pts, 128, "round5_forward_gathers_ms"),
            (("neus_point", "!neus_encode_levels"),
             "forward = neus_point_kernel alone (below the level-major crossover: all 16 levels gathered per point + SDF "
             "linear + analytic gradient + alpha + colour MLP on MFMA)", 524.0 * pts, 128, "point_major_16"),
            (("neus_encode_levels",), "neus_encode_levels_kernel (8 gathers per hashed level and point -> 16-byte records)",
             (352.0 + 8.0) * pts, 88, "level_major_hashed"),
            (("neus_point", "?neus_encode_levels"), "neus_point_kernel (streams the records; dense-level gathers, SDF layer, alpha, colour MLP)",
             (160.0 + 12.0) * pts, 40, "point_major_dense"),
            (("neus_backward_points_binned",), "neus_point_bwd_kernel<binned, aux> (streams the forward's per-level records, "
             "table-gradient records = pass 1 of bin-and-reduce)", 1036.0 * pts, 0, None),
            (("grid_bin_reduce",), "grid_bin_reduce_kernel (pass 2: exact integer LDS sums per 8192-entry bin)",
             6.0 * 88.0 * pts + nh * (1 << 19) * 4.0 * 2, 0, None),
            (("mlp_backward",), "neus_mlp_bwd_kernel (fused colour-MLP backward, MFMA)", (160.0 + 12.0 + 6.0 + 160.0) * pts, 0, None)]
    out = []
    for keys, name, nbytes, loads, leg in spec:
        absent = [k[1:] for k in keys if k.startswith("!")]             # "!k": only when kernel k did NOT run
        present = [k[1:] for k in keys if k.startswith("?")]            # "?k": only when k ran (its time is not added)
        keys = tuple(k for k in keys if k[0] not in "!?")
        ran = lambda k: k in t and t[k][1] > 0                          # noqa: E731
        if not all(ran(k) for k in keys + tuple(present)) or any(ran(k) for k in absent):
            continue
        us = sum(1e3 * t[k][0] / t[k][1] for k in keys)
        gbs = nbytes / (us * 1e-6) / 1e9
        traffic = [pmc.get(k) for k in keys]
        ent = {"kernel": name + f" @ {n_rays} rays x 72 samples", "bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS,
               "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nbytes, "kernel_avg_us": us,
               "launches_timed": t[keys[0]][1], "traffic": sum(traffic) if all(v is not None for v in traffic) else None,
               "traffic_source": "profiles/r06_pmc_neus.json (committed PMC passes)" if all(v is not None for v in traffic) else None}
        if keys[0] in ("grid_bin_reduce", "mlp_backward"):
            ent["bytes_are"] = ("this design's own traffic (record queues read + table written / saved activations + dX), NOT a "
                                "SURVEY 8(d) figure: the step's fraction on SURVEY's bytes is the `mapper step, whole` entry")
        if loads:
            ent["gather_rate_Ggathers_per_s"] = inb * float(loads) / (us * 1e-6) / 1e9
            if ceil_ and leg:
                rep_ms = ceil_[leg] if isinstance(ceil_[leg], float) else ceil_[leg]["ms"]
                ent["gather_only_replay_us"] = rep_ms * 1e3
                ent["gather_frac_of_replay"] = rep_ms * 1e3 / us
                ent["gather_ceiling_source"] = ceil_["source"]
        out.append(ent)
    step_us = 1e3 * sum(v[0] for v in t.values()) / steps
    return out, {"library_kernels_us_per_step": step_us,
                 "per_kernel_us": {k: round(1e3 * v[0] / steps, 2) for k, v in sorted(t.items(), key=lambda kv: -kv[1][0])}}


CONV_LAYERS = (("gru_zr", 320, 256), ("gru_q", 320, 128), ("heads", 128, 384), ("corr_enc2", 128, 128))


def pmc_traffic(name, key="traffic_bytes_per_launch"):
    """HBM bytes per launch from a COMMITTED rocprofv3 --pmc pass of the same launch shape (FETCH_SIZE / WRITE_SIZE in
    separate passes, corrected as MI355X_MICROARCH.md prescribes).  Not measured in this run: the source is named."""
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return {"traffic": None}
    return {"traffic": json.load(open(path)).get(key),
            "traffic_source": f"profiles/{name} (committed PMC passes, not re-measured by this run)"}


def conv_roofline(device, E, ht, wd):
    from go_slam_amd import droid_net as DN
    layers, flops_sum, ms_sum = [], 0.0, 0.0
    for name, c, o in CONV_LAYERS:
        x = torch.randn(E, c, ht, wd, device=device).half().contiguous(memory_format=torch.channels_last)
        w = (torch.randn(o, c, 3, 3, device=device) / (3.0 * c ** 0.5)).half()
        assert DN._use_own_conv3x3(x, w, 1, 1), "the update operator must be on the package's own 3x3 convolution"
        ms = time_op(lambda: DN.conv3x3_hip(x, w), iters=10, warm=3)
        fl = 2.0 * E * ht * wd * 9 * c * o
        layers.append({"layer": name, "c_in": c, "c_out": o, "us": ms * 1e3, "tflops": fl / (ms * 1e-3) / 1e12})
        flops_sum += fl
        ms_sum += ms
        del x, w
    tf = flops_sum / (ms_sum * 1e-3) / 1e12
    kern = "conv3x3_pp_kernel (two-group ping-pong implicit GEMM, LDS-DMA staging)"
    out = {"kernel": kern + ", fp16 MFMA 32x32x16 / fp32 accumulate; the update operator's four 3x3 layer shapes",
           "bound": "mfma", "achieved": tf, "peak": MFMA_F16_PEAK_TFLOPS, "unit": "TFLOP/s",
           "frac": tf / MFMA_F16_PEAK_TFLOPS, "algorithmic_flops_per_update": flops_sum,
           "kernel_us_per_update": ms_sum * 1e3, "layers": layers}
    out.update(pmc_traffic("r05_pmc_conv3x3_pp.json"))
    out["same_box_gemm_comparator"] = "profiles/r06_gemm_comparator.json: hipBLASLt at this layer's own M x N x K 0.31-0.34 of peak, 8192^3 0.50-0.53"
    return out


def summary(line):
    """The headline number of every leg, compact, as the line's last key (a driver that keeps only the tail of stdout
    still shows them)."""
    def g(d, *ks):
        for k in ks:
            d = d.get(k) if isinstance(d, dict) else None
        return round(d, 4) if isinstance(d, float) else d
    pm = {}
    for ent in line.get("roofline_other", []):
        k = ent.get("kernel", "")
        for tag, key in (("forward = ", "fwd"), ("neus_point_bwd_kernel", "bwd1"), ("grid_bin_reduce", "binred"),
                         ("neus_mlp_bwd", "mlpbwd")):
            if k.startswith(tag) and "kernel_avg_us" in ent:
                rays = "32768" if "32768" in k else "4096"
                pm[f"{key}@{rays}"] = [round(ent["kernel_avg_us"], 1), round(ent["frac"], 3),
                                       round(ent.get("gather_frac_of_replay", 0.0), 3) or None]
    return {"keyframes_per_s": g(line, "value"), "ms_per_keyframe": g(line, "ms_per_step"),
            "conv3x3_frac_of_mfma_peak": g(line, "roofline", "frac"),
            "neus_render_rays_per_s": g(line, "neus_render", "value"),
            "neus_train_32768": [g(line, "neus_train", "ms_per_step"), g(line, "neus_train", "value")],
            "neus_train_4096_per_gpu": [g(line, "neus_train_weak", "ms_per_step"), g(line, "neus_train_weak", "value")],
            "exchange_ms": [g(line, "neus_train", "exchange_exposed_ms"), g(line, "neus_train_weak", "exchange_exposed_ms")],
            "sharded_schedule_rccl_1rank_ms_[32768,4096]": [g(line, "neus_train_sharded_schedule_rccl_1rank", "32768", "ms_per_step"),
                                                             g(line, "neus_train_sharded_schedule_rccl_1rank", "4096", "ms_per_step")],
            "pathM_kernels_[us,frac_hbm,frac_replay]": pm,
            "mapper_step_frac_hbm_on_survey_bytes_[32768,4096]": [round(e["frac"], 4) for e in line.get("roofline_other", [])
                                                                 if str(e.get("kernel", "")).startswith("mapper step, whole")],
            "ba_2iter_ms": g(line, "breakdown_ms", "ba_2iter_ms"),
            "mono_window": [g(line, "mono_window", "keyframes_per_s"), g(line, "mono_window", "ba_2iter_ms")],
            "stress_step_ms": g(line, "global_ba_stress", "update_lowmem_step_ms"),
            "altcorr_[ms_per_step,frac_vector_roof]": [g(line, "global_ba_stress", "altcorr_roofline", "kernel_ms_per_step"),
                                                        g(line, "global_ba_stress", "altcorr_roofline", "frac_of_fp32_vector_roof_157TF")],
            "motion_filter_frame_ms": g(line, "breakdown_ms", "motion_filter_frame_ms"),
            "e2e_[frontend_ms_per_kf,frames_per_s,edges]": [g(line, "sequence", "frontend_e2e_ms_per_keyframe"),
                                                            g(line, "sequence", "frames_per_s"), g(line, "sequence", "edges_mean")],
            "e2e_sparse_[frontend_ms_per_kf,edges,over_unit]": [g(line, "sequence_sparse", "frontend_e2e_ms_per_keyframe"),
                                                                 g(line, "sequence_sparse", "edges_mean"), g(line, "sequence_sparse", "e2e_over_unit")],
            "e2e_loop_closure_frontend_ms_per_kf": g(line, "sequence_loop_closure", "frontend_e2e_ms_per_keyframe"),
            "e2e_shared_video_frontend_ms_per_kf": g(line, "sequence_shared_video", "frontend_e2e_ms_per_keyframe"),
            "global_ba_ms_[200kf,8steps]": g(line, "global_ba", "global_ba_ms"),
            "mapper_ms_per_joint_iteration": g(line, "mapper_call", "mapper_ms_per_joint_iteration"),
            "n_gpus": line.get("n_gpus"), "rccl_ranks": line.get("rccl_ranks")}


def sharded_schedule_one_rank(device):
    """The world > 1 schedule of the mapper step on THIS one GPU, in an RCCL (`nccl`) process group of one rank: global
    counts outside the graph, two hipGraphs around the early reduce-scatter, fp16 reduce-scatter, slice AdamW, deferred
    in-place all-gather -- the collectives are RCCL's own one-rank copies on its stream.  What it measures: the fixed cost
    of that schedule against the whole-step graph of `neus_train` / `neus_train_weak` (no inter-GPU transfer is involved:
    the exchange itself stays unmeasured until an N > 1 run)."""
    import torch.distributed as dist
    created = False
    if not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(free_port()))
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
        created = True
        flush_c_stdio()
    try:
        out = {"collective_backend": dist.get_backend(), "ranks": dist.get_world_size(),
               "collective_selftest_ok": collectives_selftest(device, 0, 1)[0]}
        for rays in (32768, 4096):
            r = neus_train_bench(device, 0, 1, global_rays=rays, sharded=True)
            out[str(rays)] = {"ms_per_step": r["ms_per_step"], "graph_replay": r["graph_replay"],
                              "collectives_captured_in_the_graph": r.get("collectives_captured"),
                              "capture_error": r.get("capture_error"),
                              "exchange_per_rank": r["exchange_per_rank"], "final_loss": r["final_loss"]}
    finally:
        if created:
            dist.destroy_process_group()
        flush_c_stdio()
    return out


def collectives_selftest(device, rank, world):
    """The collective calls of the sharded mapper step (go_slam_amd/neus/distributed.py: in-place fp16 reduce-scatter,
    asynchronous in-place fp16 all-gather, fp32 all-reduce) on 8-element slices, checked against their closed forms; the
    ranks agree on the result with one MIN all-reduce (the call the launch evidence above has already exercised)."""
    import torch.distributed as dist
    from go_slam_amd.neus import distributed as D
    ok, err = 1.0, ""
    sync = torch.cuda.synchronize if torch.device(device).type == "cuda" else (lambda: None)
    try:
        full = (torch.arange(8 * world, device=device, dtype=torch.float32) % 7).to(torch.float16)
        want = full.float() * world
        out = torch.empty(8, device=device, dtype=torch.float16)
        fin = D.reduce_scatter_sum_(out, full.clone(), async_op=True)
        fin()
        sync()
        if not torch.equal(out.float(), want[8 * rank:8 * rank + 8]):
            raise RuntimeError("reduce-scatter result differs")
        buf = torch.zeros(8 * world, device=device, dtype=torch.float16)
        buf[8 * rank:8 * rank + 8] = float(rank + 1)
        fin = D.all_gather_into_(buf, buf[8 * rank:8 * rank + 8], async_op=True)
        fin()
        sync()
        if not torch.equal(buf.float().reshape(world, 8)[:, 0].cpu(), torch.arange(1, world + 1, dtype=torch.float32)):
            raise RuntimeError("all-gather result differs")
        t = torch.full((3,), float(rank), device=device)
        D.all_reduce_sum_(t)
        if float(t[0]) != world * (world - 1) / 2:
            raise RuntimeError("all-reduce result differs")
    except Exception as exc:
        ok, err = 0.0, repr(exc)
    flag = torch.tensor([ok], device=device)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() >= 1.0), err


def flush_c_stdio():
    """RCCL prints a banner ('Librccl path : ...') through C stdio; behind a pipe that buffer is only flushed at process
    exit, i.e. AFTER Python's own prints -- the bench line must stay the last line of stdout, so the C buffers are
    flushed as soon as a process group exists and again before the line is printed."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def self_launch(n, argv):
    """`python bench.py --gpus N` without a launcher: one rank per GPU under torch.distributed.run on this node."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + argv
    return subprocess.run(cmd, env=env).returncode


def launch_only(world, rank):
    """GS_BENCH_LAUNCH_ONLY=1: rendezvous check of the N > 1 launch path without touching a GPU (CPU test): every rank
    joins a gloo group, all-reduces a one, rank 0 prints what the real run would put in n_gpus / rccl_ranks."""
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if world > 1:
        dist.init_process_group("gloo")
        t = torch.ones(1)
        dist.all_reduce(t)
        ranks = int(t.item())
        dist.destroy_process_group()
    else:
        ranks = 1
    if rank == 0:
        print(json.dumps({"launch_only": True, "n_gpus": world, "rccl_ranks": ranks}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s); n_gpus reports the ranks that "
              "actually run", file=sys.stderr)
    distributed = world > 1
    if os.environ.get("GS_BENCH_LAUNCH_ONLY") == "1":
        return launch_only(world, rank)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (there is no CPU path)"
    # GS_BENCH_SMOKE_ONE_GPU=1: functional smoke test of the N>1 code path on a 1-GPU box (all ranks
    # on cuda:0, gloo instead of RCCL -- RCCL refuses two ranks on one device).  Never used for numbers.
    smoke_one_gpu = os.environ.get("GS_BENCH_SMOKE_ONE_GPU") == "1"
    if smoke_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if smoke_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)

    rccl_ranks, backend = 1, "none"
    if distributed:         # evidence that N ranks really exchange data on their devices
        t = torch.ones(1, device=device)
        dist.all_reduce(t)
        rccl_ranks, backend = int(t.item()), dist.get_backend()
        flush_c_stdio()

    video, update_op, graph, _ = build_state(device)

    def barrier():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        keyframe_step(graph)
    # What a long-running tracker process does once after start-up: move the module / tensor object graph built so
    # far out of the cyclic collector's reach.  Without it CPython's first full collection lands around the 10th
    # keyframe and stalls the launch thread for ~23 ms (tools/debug_bench_gap.py: 15.2 ms per keyframe before and
    # after, one 38.5 ms keyframe in between); garbage created from here on is still collected.
    import gc
    gc.collect()
    gc.freeze()
    barrier()
    tic = time.perf_counter()
    for _ in range(args.steps):
        keyframe_step(graph)
    barrier()
    elapsed = time.perf_counter() - tic
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    finite = bool(torch.isfinite(video.poses).all()) and bool(torch.isfinite(video.disps).all())

    ms_per_step = 1e3 * elapsed / args.steps
    value = world * args.steps / elapsed          # aggregate keyframes/s over all replicas
    line = {
        "metric": "BA keyframes/s (6 x FactorGraph.update: reproject + corr lookup + GRU + 2 GN BA iters) on synthetic 640x480 RGB-D",
        "value": value, "unit": "keyframes/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (BA/geometry; fp64 solve), f16 (corr volume, GRU convs)", "data": "synthetic",
        "config": {"workload": "configs[1]-shaped frontend window: 60x80 maps (480x640 RGB-D), P=25 keyframes, "
                               "E=75 edges, 6 updates/keyframe, iters=2, RGB-D depth prior, upsample on",
                   "parallelism": f"replicas x{world} (tracking does not shard)"},
        "updates_per_s": value * UPDATES_PER_KF, "state_finite": finite,
        "rccl_ranks": rccl_ranks, "collective_backend": backend,
    }
    # collective legs: every rank takes part.  With N > 1 the exact collective calls of the sharded mapper step are first
    # issued on small tensors and the ranks agree on the outcome -- a launcher / RCCL problem in that path must cost the
    # path-M legs of the line, never the tracking headline above (tracking does not communicate).
    coll_ok, coll_err = collectives_selftest(device, rank, world) if distributed else (True, "")

    def train_leg(**kw):
        if not coll_ok:
            return {"error": "collective self-test failed on at least one rank: " + (coll_err or "(another rank)")}
        try:
            return neus_train_bench(device, rank, world, **kw)
        except Exception as exc:                       # (a one-sided failure would hang the peers in their next collective;
            return {"error": repr(exc)}                #  symmetric ones -- API / shape / backend refusals -- end here)
    train = train_leg()
    train_weak = train_leg(global_rays=4096 * world, scaling="weak")
    if rank == 0:
        line["neus_train"] = train
        line["neus_train_weak"] = train_weak
        br = op_breakdown(video, update_op, graph)
        line["breakdown_ms"] = {k: round(v, 4) for k, v in br.items() if k.endswith("_ms")}
        line["ba_gn_iters_per_s"] = 2.0 / (br["ba_2iter_ms"] * 1e-3)
        if world == 1:      # single-GPU measurement (tracking does not shard); keeps the N > 1 runs short
            try:
                line["global_ba_stress"] = global_ba_stress(device)
            except Exception as exc:                   # an auxiliary leg must not cost the bench line
                line["global_ba_stress"] = {"error": repr(exc)}
        ht, wd = graph.ht, graph.wd
        # ---- roofline: the time-dominant hand-written kernel = the update operator's 3x3 convolutions (about half of
        # an update's GPU time, profiles/r02_tracking_kernel_stats.md).  MFMA-bound; achieved = algorithmic FLOPs
        # (2 * E * h * w * 9 * C_in * C_out, no tile padding counted) / launch time, measured here with events on the
        # launch stream for each of the four layer shapes an update issues, and aggregated over them.
        line["roofline"] = conv_roofline(device, int(br["edges"]), ht, wd)
        algo_bytes = 912.0 * ht * wd * br["edges"]            # 912*HW B per edge per lookup
        t_s = br["corr_lookup_ms"] * 1e-3
        achieved = algo_bytes / t_s / 1e9
        # 2.656 HW^2 bytes per edge (inputs + the four levels written once); the tile8 layout this launch
        # writes pads levels 0-1 to 8x8 tiles (+6.7 % at 60x80) -- the padded count is reported next to it, not used
        vol_algo = 8 * (2 * 128 * ht * wd * 2 + 2.0 * (ht * wd) ** 2 * (1 + 0.25 + 1 / 16 + 1 / 64))
        bgb = vol_algo / (br["corr_build_8edges_ms"] * 1e-3) / 1e9
        line["roofline_other"] = []
        if "corr_lookup_enc0_ms" in br:      # the production lookup: fused with corr_encoder[0]
            fb = (4 * 64 * 2 + 8 + 128 * 2) * ht * wd * br["edges"]       # window reads + coords + 128 fp16 outputs per pixel
            fa = fb / (br["corr_lookup_enc0_ms"] * 1e-3) / 1e9
            line["roofline_other"].append(
                {"kernel": "corr_lookup_enc_kernel<tile8> (cooperative 4-level lookup + corr_encoder[0] 196->128 on MFMA, "
                           "bias + ReLU; the 196-channel features never reach HBM)", "bound": "hbm", "achieved": fa,
                 "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fa / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": fb,
                 "kernel_avg_us": br["corr_lookup_enc0_ms"] * 1e3,
                 "replaces_us": (br["corr_lookup_ms"] * 1e3, "+ conv1x1 196->128"),
                 "note": "not HBM-bound in the roofline sense: the 8x8 windows over-fetch 2.4x (traffic 1.93x algorithmic) and "
                         "the layout model (profiles/r03_lookup_layout_model.json) finds at most -21 % from re-tiling: 0.28 is "
                         "the floor of this design", **pmc_traffic("r03_pmc_corr_lookup.json")})
        line["roofline_other"] += [
            dict({"kernel": "corr_pyramid_coop_kernel<tile8> (fp16 NHWC, wave-cooperative fused 4-level lookup; the "
                            "unfused ABI entry)",
                  "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": algo_bytes,
                  "kernel_avg_us": br["corr_lookup_ms"] * 1e3},
                 **pmc_traffic("r03_pmc_corr_lookup.json", "traffic_bytes_per_launch_unfused")),
            dict({"kernel": "corr_volume_kernel (MFMA all-pairs volume + 3 pooled levels, written once)", "bound": "hbm",
                  "achieved": bgb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": bgb / HBM_PEAK_GBS,
                  "algorithmic_bytes_per_launch": vol_algo, "bytes_written_with_tile8_padding": br["corr_build_bytes"],
                  "kernel_avg_us": br["corr_build_8edges_ms"] * 1e3,
                  "note": "8 edges per launch (one keyframe's new factors); the time includes the operand re-ordering "
                          "and level-3 pooling launches; HBM traffic = algorithmic; a tile's life is dominated by its "
                          "operand loads queueing behind the CU's stores (phase timeline in DESIGN 3)"},
                 **pmc_traffic("r03_pmc_corr_volume.json"))]
        try:        # the frame encoders' elementwise tail (gs_norm_act) at layer1's shape: never fatal for the headline
            line["roofline_other"].append(encoder_tail_roofline(device))
        except Exception as exc:
            line["roofline_other"].append({"kernel": "gs_norm_act", "error": repr(exc)})
        line["neus_render"] = neus_render_bench(device)
        line["roofline_other"].append(dict(line["neus_render"]["mlp_mfma"], bound="mfma"))
        if world == 1:
            for nr in (4096, 32768):        # the kernels that own path M, timed per kernel (eager step + the library's timer)
                try:
                    ents, summ = neus_kernel_rooflines(device, nr)
                    line["roofline_other"] += ents
                    line.setdefault("neus_step_kernels", {})[str(nr)] = summ
                except Exception as exc:
                    line["roofline_other"].append({"kernel": f"path M kernels @ {nr}", "error": repr(exc)})
            try:
                line["mono_window"] = mono_window(device)
            except Exception as exc:
                line["mono_window"] = {"error": repr(exc)}
            # ---- end to end, steady state (north_star: "throughput on synthetic 640x480 RGB-D sequences"): outside the
            # headline's timed region, own keys
            for key, kw in (("sequence", {}), ("sequence_sparse", {"step_m": 0.02, "step_deg": 0.6, "keyframes": 24}),
                            ("sequence_loop_closure", {"enable_loop": True, "keyframes": 24}),
                            ("sequence_shared_video", {"shared_video": True, "keyframes": 24})):
                try:
                    line[key] = sequence_bench(device, **kw)
                except Exception as exc:
                    line[key] = {"error": repr(exc)}
            try:
                line["global_ba"] = backend_bench(device)
            except Exception as exc:
                line["global_ba"] = {"error": repr(exc)}
            try:
                line["mapper_call"] = mapper_call_bench(device)
            except Exception as exc:
                line["mapper_call"] = {"error": repr(exc)}
            try:        # the sharded (world > 1) schedule of the mapper step under RCCL, one rank: its fixed cost on hardware
                line["neus_train_sharded_schedule_rccl_1rank"] = sharded_schedule_one_rank(device)
            except Exception as exc:
                line["neus_train_sharded_schedule_rccl_1rank"] = {"error": repr(exc)}
        # the mapper step as a WHOLE on SURVEY 8(d)'s bytes: render 524 B + train 1036 B (512 gathers + 512 scatter + 12 in) per
        # sample point + AdamW's 353 MB (12.6 M parameters x 28 B), over the graph-replayed step time of the legs above
        for leg, rays in (("neus_train", 32768), ("neus_train_weak", 4096)):
            ms = (line.get(leg) or {}).get("ms_per_step")
            if world == 1 and ms:
                nb = (524.0 + 1036.0) * rays * 72 + 353.0e6
                line["roofline_other"].append(
                    {"kernel": f"mapper step, whole (sample + forward + loss + backward + bin reduce + clip + AdamW, one hipGraph) @ {rays} rays x 72 samples",
                     "bound": "hbm", "achieved": nb / (ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": nb / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "algorithmic_bytes_per_launch": nb, "kernel_avg_us": ms * 1e3,
                     "note": "SURVEY 8(d): (524 + 1036) B per sample point + 353 MB of optimiser traffic"})
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(2)
            try:        # path M on the host cores (north_star: the render batches "alongside the reference's CPU path")
                line["cpu_baseline"].update(cpu_baseline_neus())
            except Exception as exc:
                line["cpu_baseline"]["neus_error"] = repr(exc)
        st = line.get("global_ba_stress") or {}
        if "altcorr_roofline" in st:
            line["roofline_other"].append(st["altcorr_roofline"])
        line["summary"] = summary(line)         # LAST key: a reader of the line's tail sees every leg's headline number
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    flush_c_stdio()                             # every C-level byte of this process is out before the line is printed:
    if rank == 0:                               # the line stays the LAST line of stdout
        sys.stdout.flush()
        print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
