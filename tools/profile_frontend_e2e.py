"""The tracker end to end as a stand-alone command for rocprofv3 (kernel trace) and for the torch-op tracer:
    python tools/profile_frontend_e2e.py [keyframes] [--ops] [--loop]
bench.py's `sequence_bench` (MotionFilter.track per frame + Frontend.__call__ per keyframe on a synthetic 640x480 RGB-D
sequence, steady state), then a marker launch and `keyframes` more keyframes: tools/summarize_kernels.py --after erfinv
reports them per keyframe.  --ops: instead of running under rocprofv3, count the torch operators, host syncs (`.item()`,
`.cpu()`, `.tolist()`, nonzero) and library launches the Python callers issue per keyframe -- the tracer that found round
5's caching faults -- and print them as JSON."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402

args = [a for a in sys.argv[1:] if not a.startswith("--")]
n_kf = int(args[0]) if args else 16
dev = torch.device("cuda:0")
out, (net, video, fe, mf, frame_stage) = bench.sequence_bench(dev, keyframes=8, warm_keyframes=34,
                                                               enable_loop="--loop" in sys.argv, return_state=True,
                                                               spare_keyframes=4 * n_kf + 8, freeze_gc=True)
torch.cuda.synchronize()

if "--ops" in sys.argv:
    from collections import Counter
    from torch.utils._python_dispatch import TorchDispatchMode

    class Tracer(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.ops = Counter()

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            self.ops[str(func.name())] += 1
            return func(*args, **(kwargs or {}))
    from go_slam_amd import _lib
    import contextlib
    res = {}
    for stage in ("frontend", "frames"):
        tr = Tracer()
        launches = Counter()
        for _ in range(n_kf):
            with (tr if stage == "frames" else contextlib.nullcontext()), _lib.kernel_timer(dev) as kt:
                frame_stage()
                torch.cuda.synchronize()
            if stage == "frames":
                launches.update({k: v[1] for k, v in kt.read().items()})
            with (tr if stage == "frontend" else contextlib.nullcontext()), _lib.kernel_timer(dev) as kt:
                fe()
                torch.cuda.synchronize()
            if stage == "frontend":
                launches.update({k: v[1] for k, v in kt.read().items()})
        sync_ops = ("aten::_local_scalar_dense", "aten::nonzero", "aten::item")
        host_copies = sum(v for k, v in tr.ops.items() if k == "aten::_to_copy")
        res[stage] = {"torch_ops_per_keyframe": round(sum(tr.ops.values()) / n_kf, 1),
                      "host_sync_ops_per_keyframe": {k: round(v / n_kf, 2) for k, v in tr.ops.items() if k in sync_ops},
                      "to_copy_per_keyframe": round(host_copies / n_kf, 2),
                      "library_launches_per_keyframe": round(sum(launches.values()) / n_kf, 1),
                      "library_launches": {k: round(v / n_kf, 2) for k, v in launches.most_common(40)},
                      "top_ops": {k: round(v / n_kf, 1) for k, v in tr.ops.most_common(30)}}
    print(json.dumps(res, indent=1))
    sys.exit(0)

if "--copies" in sys.argv:                           # WHERE the host <-> device copies and scalar reads of a keyframe come from
    import traceback
    from collections import Counter
    from torch.utils._python_dispatch import TorchDispatchMode

    class CopyTracer(TorchDispatchMode):
        def __init__(self):
            super().__init__()
            self.sites = Counter()

        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            name = str(func.name())
            kind = None
            if name == "aten::_to_copy" and torch.is_tensor(args[0]):
                dst = (kwargs or {}).get("device")
                if dst is not None and torch.device(dst).type != args[0].device.type:
                    kind = "H2D" if args[0].device.type == "cpu" else "D2H"
            elif name in ("aten::_local_scalar_dense", "aten::nonzero", "aten::item"):
                kind = name.split("::")[1]
            elif name in ("aten::lift_fresh", "aten::scalar_tensor"):
                kind = None
            if kind is not None:
                fr = [f for f in traceback.extract_stack() if "go_slam_amd" in f.filename or "bench.py" in f.filename]
                site = " < ".join(f"{os.path.basename(f.filename)}:{f.lineno}" for f in fr[-3:][::-1])
                self.sites[(kind, site)] += 1
            return func(*args, **(kwargs or {}))
    tr = CopyTracer()
    for _ in range(n_kf):
        if "--frames" in sys.argv:                   # the four MotionFilter.track calls instead of the frontend
            with tr:
                frame_stage()
        else:
            frame_stage()
        torch.cuda.synchronize()
        if "--frames" in sys.argv:
            fe()
        else:
            with tr:
                fe()
        torch.cuda.synchronize()
    print(json.dumps([[k[0], k[1], round(v / n_kf, 2)] for k, v in tr.sites.most_common(60)], indent=0))
    sys.exit(0)

if "--hostgaps" in sys.argv:                         # host time between consecutive library launches of a keyframe (> 60 us)
    from collections import defaultdict
    from go_slam_amd import _lib
    L = _lib.lib()
    log = []
    real = {}

    def wrap(name):
        fn = getattr(L, name)
        real[name] = fn

        def call(*a, **k):
            log.append((time.perf_counter(), name))
            return fn(*a, **k)
        call.restype, call.argtypes = fn.restype, fn.argtypes
        return call
    for name in list(_lib.SIGNATURES):
        if name.startswith("gs_") and "workspace" not in name and "blocks" not in name and "timing" not in name:
            try:
                setattr(L, name, wrap(name))
            except Exception:
                pass
    agg = defaultdict(lambda: [0, 0.0])
    for _ in range(n_kf):
        frame_stage()
        torch.cuda.synchronize()
        log.clear()
        fe()
        torch.cuda.synchronize()
        for (t0, a), (t1, b) in zip(log, log[1:]):
            if t1 - t0 > 60e-6:
                e = agg[(a, b)]
                e[0] += 1
                e[1] += (t1 - t0) * 1e6
    print(json.dumps([[k[0], k[1], round(v[0] / n_kf, 2), round(v[1] / v[0])] for k, v in
                      sorted(agg.items(), key=lambda kv: -kv[1][1])[:25]], indent=0))
    sys.exit(0)

if "--cprofile" in sys.argv:                         # where the HOST time of a keyframe goes (no device sync inside)
    import cProfile
    import io
    import pstats
    pr = cProfile.Profile()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_kf):
        frame_stage()
        pr.enable()
        fe()
        pr.disable()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    buf = io.StringIO()
    pstats.Stats(pr, stream=buf).sort_stats("cumulative").print_stats(45)
    print(buf.getvalue())
    print("wall ms per keyframe (frames + frontend):", round(1e3 * wall / n_kf, 3))
    sys.exit(0)

torch.erfinv(torch.zeros(1, device=dev))            # marker launch
t0 = time.perf_counter()
for _ in range(n_kf):
    frame_stage()
    fe()
torch.cuda.synchronize()
print("done", n_kf, "keyframes", round(1e3 * (time.perf_counter() - t0) / n_kf, 3), "ms per keyframe (frames + frontend)",
      json.dumps({k: out[k] for k in ("frontend_e2e_ms_per_keyframe", "motion_filter_ms_per_frame", "edges_mean")}))
