"""Gather-only replay of the hash grid's OWN index stream on the bench's ray batches -> the ceiling bench.py quotes
`neus_point_kernel` / `neus_encode_levels_kernel` against (profiles/r05_gather_replay.json).  See tools/gather_replay.hip.

    python tools/gather_replay.py [out.json]          (needs an MI355X; builds tools/gather_replay.so when it is missing)

Per batch size (4096 and 32768 rays x 72 samples, the batches of bench.py's neus_render / neus_train legs, same seed):
the in-bound sample points' 16 x 8 table indices are written once by the production index arithmetic, then replayed
  point_major_16          every level per lane, level by level            (the access order of the round-4 forward)
  level_major_hashed      (level, chunk) items, XCD-consecutive           (neus_encode_levels_kernel, round 5)
  point_major_dense       levels 0..4 per lane                            (what neus_point_kernel still gathers itself)
Rates count ONLY the table loads (8 per level and in-bound point)."""
import ctypes
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from go_slam_amd import _lib  # noqa: E402
import go_slam_amd.neus as neus  # noqa: E402

SO = os.path.join(ROOT, "tools", "gather_replay.so")


def load():
    if not os.path.exists(SO):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-shared", "-fPIC",
                               os.path.join(ROOT, "tools", "gather_replay.hip"), "-o", SO])
    return ctypes.CDLL(SO)


def batch(n, device, seed=43):
    g = torch.Generator().manual_seed(seed)
    model = neus.InstantNeuS({}, [[-5.0, 5.0]] * 3).to(device)
    with torch.no_grad():
        model.sdf_network.encoding.encoding.params.copy_(
            (torch.rand(model.sdf_network.encoding.encoding.params.shape, generator=g) - 0.5) * 0.02)
    o = (torch.rand(n, 3, generator=g) * 6 - 3).to(device)
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1).to(device)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[torch.rand(n, generator=g) < 0.1] = 0
    pr = torch.rand(24, generator=g).to(device)
    z, dist = neus.Renderer(N_samples=24, N_surface=48).sample(o, d, model.bound, gt.to(device), pr)
    pts = o[:, None] + d[:, None] * (z + dist / 2)[..., None]
    b = model.bound
    mask = ((pts > model.realtime_bound[:, 0]) & (pts < model.realtime_bound[:, 1])).all(-1)
    q = ((pts - b[:, 0]) / (b[:, 1] - b[:, 0]) * 2 - 1).clamp(-1, 1)
    return model, ((q + 1) / 2).reshape(-1, 3).contiguous(), mask.reshape(-1).to(torch.uint8).contiguous()


def timed(fn, reps=10):
    fn()
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    b.synchronize()
    return a.elapsed_time(b) / reps


def main():
    dev = torch.device("cuda:0")
    G = load()
    meta = _lib.grid_meta()
    first_hashed = min(l for l in range(16) if meta.hashed[l])
    res = {"device": torch.cuda.get_device_name(0), "first_hashed_level": first_hashed, "batches": {}}
    for n in (4096, 32768):
        model, view, mask = batch(n, dev)
        np_ = view.shape[0]
        tab = model.sdf_network.encoding.encoding.params_half().view(torch.int32)
        idx = torch.empty(16 * 8 * np_, dtype=torch.int32, device=dev)
        out = torch.empty(16 * np_, dtype=torch.int32, device=dev)
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())   # noqa: E731
        assert G.gr_index_stream(P(view), P(mask), P(idx), np_, ctypes.byref(meta), st) == 0
        torch.cuda.synchronize()
        inb = int(mask.sum())
        t_pm = timed(lambda: G.gr_replay_point_major(P(idx), P(tab), np_, 0, 16, P(out), st))
        t_lm = timed(lambda: G.gr_replay_level_major(P(idx), P(tab), np_, first_hashed, P(out), st))
        t_pd = timed(lambda: G.gr_replay_point_major(P(idx), P(tab), np_, 0, first_hashed, P(out), st))
        nh = 16 - first_hashed
        gat = lambda levels: inb * 8.0 * levels   # noqa: E731
        res["batches"][str(n)] = {
            "points": np_, "points_in_bound": inb,
            "point_major_16": {"ms": t_pm, "Ggathers_per_s": gat(16) / t_pm / 1e6},
            "level_major_hashed": {"levels": nh, "ms": t_lm, "Ggathers_per_s": gat(nh) / t_lm / 1e6},
            "point_major_dense": {"levels": first_hashed, "ms": t_pd, "Ggathers_per_s": gat(first_hashed) / t_pd / 1e6},
            "round5_forward_gathers_ms": t_lm + t_pd}
        del idx, out
    txt = json.dumps(res, indent=1)
    print(txt)
    if len(sys.argv) > 1:
        os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
        open(sys.argv[1], "w").write(txt)


if __name__ == "__main__":
    main()
