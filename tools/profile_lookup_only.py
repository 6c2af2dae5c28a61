"""The correlation lookup of the bench workload as a stand-alone command for rocprofv3 --pmc passes: 5 launches of the
unfused cooperative kernel (gs_corr_lookup_pyramid) and 5 of the production kernel fused with corr_encoder[0]
(gs_corr_lookup_enc)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402

dev = torch.device("cuda:0")
video, op, graph, _ = bench.build_state(dev)
coords1, _ = video.reproject(graph.ii, graph.jj)
wpad, bias = op._corr_enc0_padded()
for _ in range(5):
    graph.corr(coords1)
    graph.corr.lookup_encoded(coords1, wpad, bias)
torch.cuda.synchronize()
