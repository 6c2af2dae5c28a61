"""MotionFilter.track on one 480x640 RGB-D input frame (bench.motion_filter_frame_ms's workload) as a stand-alone command
for rocprofv3 --kernel-trace:   python tools/profile_motion_filter.py [frames]
Every frame does the same work; per-frame numbers are trace totals / frames after the marker launch
(tools/summarize_kernels.py --steps N --after erfinv: model construction, the first keyframe and the warm frames are not counted)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 10
frame = bench.motion_filter_frame_ms(torch.device("cuda:0"), return_fn=True)
for _ in range(3):                                  # warm (lazy library state, caches)
    frame()
torch.ones(8, device="cuda:0").erfinv_()            # marker launch: summarize_kernels.py --after erfinv counts from here
for _ in range(frames):
    frame()
torch.cuda.synchronize()
print("done", frames)
