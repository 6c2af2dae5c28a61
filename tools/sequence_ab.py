"""bench.py's end-to-end sequence legs as a stand-alone command (same-box A/B runs of a knob):
    python tools/sequence_ab.py [plain] [loop] [sparse]
Prints one JSON line per leg: frontend ms per keyframe, MotionFilter ms per frame, frames per second, mean edges."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402

legs = [a for a in sys.argv[1:] if not a.startswith("--")] or ["plain", "loop"]
kw = {"plain": {"keyframes": 24}, "loop": {"enable_loop": True, "keyframes": 24},
      "sparse": {"step_m": 0.02, "step_deg": 0.6, "keyframes": 24}}
dev = torch.device("cuda:0")
for leg in legs:
    out = bench.sequence_bench(dev, **kw[leg])
    keep = {k: out[k] for k in out if k in ("frontend_e2e_ms_per_keyframe", "motion_filter_ms_per_frame", "frames_per_s",
                                             "edges_mean", "six_update_unit_ms_on_the_final_graph", "e2e_over_unit")}
    print(json.dumps({"leg": leg, "batch_uploads": os.environ.get("GOSLAM_BATCH_UPLOADS", "1"), **keep}))
