"""A/B of the update operator's 3x3-convolution implementation on the bench workload: ms per keyframe
(bench.UPDATES_PER_KF x FactorGraph.update, as bench.py times it) with CONV3X3_IMPL = "miopen" (library convolutions, the referee setting) / "own" (gs_conv3x3_pp).  Prints one JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import go_slam_amd.droid_net as DN  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.backends.cudnn.benchmark = True
    video, op, graph, _ = bench.build_state(dev, seed=43)
    poses0, disps0 = video.poses.clone(), video.disps.clone()
    out = {}
    for impl in ("miopen", "own", "miopen", "own"):
        DN.CONV3X3_IMPL = impl
        video.poses.copy_(poses0)
        video.disps.copy_(disps0)
        ms = bench.time_op(lambda: bench.keyframe_step(graph), iters=8, warm=3)
        out.setdefault(impl, []).append(round(ms, 3))
    out["kf_per_s"] = {k: round(1000.0 / min(v), 2) for k, v in out.items()}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
