"""Cross-process use of the hot path, as src/slam.py:373-390 runs it: one DepthVideo(cfg, args) whose buffers are
`share_memory_()` HIP tensors, handed to `torch.multiprocessing` (spawn) workers that call `droid_backends.ba` on the
SAME poses / disps under different locks (frontend: factor_graph.py:246-247 under the counter lock; backend:
backend.py:112 under ba_lock['dense']).  SURVEY section 5 race row / 8b: the library is stateless per call and its
workspaces are per process, so two processes may run BA concurrently on shared buffers.

Two workers optimise DISJOINT windows of one shared video concurrently (so the result is order-independent) and the
parent compares the shared state with the CPU oracle applied serially."""
import types

import pytest
import torch
import torch.multiprocessing as mp

from go_slam_amd import synth

pytestmark = pytest.mark.gpu

SHAPE, HALF, EDGES = "tiny", 12, 30


def _cfg():
    ht, wd, _ = synth.SHAPES[SHAPE]
    return {"mode": "rgbd", "cam": {"H_out": 8 * ht, "W_out": 8 * wd}, "tracking": {"buffer": 2 * HALF + 4}}


def _problems(O):
    """Two BA problems over keyframes [0,12) and [12,24) of one 24-keyframe video."""
    vid = synth.make_video(2 * HALF, SHAPE, seed=91, rgbd=True, buffer=2 * HALF + 4)
    ht, wd, _ = synth.SHAPES[SHAPE]
    probs = []
    for k in range(2):
        ii, jj = synth.make_graph(HALF, EDGES, seed=92 + k)
        ii, jj = ii + k * HALF, jj + k * HALF
        g = torch.Generator().manual_seed(93 + k)
        c, _ = O.reproject(vid["poses"], vid["disps"], vid["intrinsics"], ii, jj)
        target = (c[0] + 0.5 * torch.randn(len(ii), ht, wd, 2, generator=g)).permute(0, 3, 1, 2).contiguous()
        weight = torch.rand(len(ii), 2, ht, wd, generator=g)
        t0, t1 = k * HALF + 1, (k + 1) * HALF
        kx = torch.unique(torch.cat([torch.arange(t0, t1), ii]))
        eta = 1e-2 * torch.rand(len(kx), ht, wd, generator=g) + 1e-4
        probs.append(dict(ii=ii, jj=jj, target=target, weight=weight, eta=eta, t0=t0, t1=t1))
    return vid, probs


def _worker(rank, video, prob, lock_kind, barrier, out_q):
    """Runs in a spawned process: the video object arrives through pickling (HIP IPC handles + shared Values)."""
    try:
        import torch
        from go_slam_amd import droid_backends as db
        dev = video.poses.device
        torch.cuda.set_device(dev)
        args = [prob[k].to(dev) for k in ("target", "weight", "eta", "ii", "jj")]
        lock = video.get_lock() if lock_kind == "counter" else video.get_ba_lock(lock_kind)
        barrier.wait(timeout=60)                   # both workers enter BA together
        for _ in range(3):                         # a few rounds each, interleaving with the other process
            with lock:
                dx, dz = db.ba(video.poses, video.disps, video.intrinsics[0].contiguous(), video.disps_sens, *args,
                               prob["t0"], prob["t1"], 2, 1e-4, 0.1, False)
                torch.cuda.synchronize()
        with video.get_lock():
            video.counter.value += HALF
        out_q.put((rank, "ok", dx.cpu().numpy()))    # a NumPy array: a torch tensor would travel as a file
                                                      # descriptor owned by this (soon finished) process
    except Exception as exc:                       # surface the failure in the parent instead of hanging it
        import traceback
        out_q.put((rank, "error", traceback.format_exc() + repr(exc)))


@pytest.mark.timeout(300)
def test_two_processes_run_ba_on_one_shared_video(built_lib):
    from go_slam_amd.depth_video import DepthVideo
    from oracle import droid_oracle as O
    vid, probs = _problems(O)
    video = DepthVideo(_cfg(), types.SimpleNamespace(device="cuda:0"))
    assert video.poses.is_cuda and video.ht == 8 * synth.SHAPES[SHAPE][0]
    video.poses.copy_(vid["poses"]); video.disps.copy_(vid["disps"])
    video.disps_sens.copy_(vid["disps_sens"]); video.intrinsics.copy_(vid["intrinsics"])
    torch.cuda.synchronize()

    ctx = mp.get_context("spawn")
    barrier, out_q = ctx.Barrier(2), ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(k, video, probs[k], ("counter", "dense")[k], barrier, out_q))
             for k in range(2)]
    for p in procs:
        p.start()
    results = {}
    try:
        for _ in range(2):
            rank, status, payload = out_q.get(timeout=240)
            assert status == "ok", payload
            results[rank] = payload
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert video.counter.value == 2 * HALF, "both workers bumped the shared keyframe counter"

    # serial oracle: 3 x (2 GN iterations) on each window
    po, do = vid["poses"].clone(), vid["disps"].clone()
    K = vid["intrinsics"][0].contiguous()
    last = {}
    for k, pr in enumerate(probs):
        for _ in range(3):
            last[k] = O.ba(po, do, K, vid["disps_sens"], pr["target"], pr["weight"], pr["eta"], pr["ii"], pr["jj"],
                           pr["t0"], pr["t1"], 2, 1e-4, 0.1, False)[0]
    assert float((po - vid["poses"]).abs().max()) > 1e-4
    torch.testing.assert_close(video.poses.cpu(), po, rtol=0, atol=2e-5)
    torch.testing.assert_close(video.disps.cpu(), do, rtol=0, atol=2e-5)
    for k in range(2):
        torch.testing.assert_close(torch.from_numpy(results[k]), last[k], rtol=2e-3, atol=2e-6)
