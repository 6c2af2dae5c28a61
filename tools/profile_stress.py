"""BASELINE configs[3]'s stress step as a stand-alone command for rocprofv3 (kernel trace):
    python tools/profile_stress.py [steps]
ONE backend invocation FactorGraph.update_lowmem(steps=`steps`, iters=2) on 200 keyframes / 1200 edges at 30x40 maps
(bench.py's `global_ba_stress`: alt-corr lookups in chunks of 13 source keyframes + update operator + a dense BA over all
edges, 6P = 1194 unknowns -> the multi-kernel blocked Cholesky), per-edge caches cold as after the frontend added
keyframes: the first step computes every chunk's hoisted context terms, the others reuse them.  Per-step numbers are
trace totals / steps (tools/summarize_kernels.py), i.e. the invocation's average step."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench                                        # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device("cuda:0")
video, update_op, graph, _ = bench.build_state(dev, seed=47, num_kf=200, num_edges=1200, shape="Scan", corr_impl="alt",
                                               upsample=False)
update_op.drop_edge_caches()
torch.cuda.synchronize()
torch.erfinv(torch.zeros(1, device=dev))            # marker launch: summarize_kernels.py --after erfinv drops the set-up
graph.update_lowmem(steps=steps, iters=2)
torch.cuda.synchronize()
print("done", steps, bool(torch.isfinite(video.poses).all()))
