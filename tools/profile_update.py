"""torch.profiler table of FactorGraph.update (bench workload) -- which kernels own an update."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
video, op, graph, _ = bench.build_state(dev)
for _ in range(8):
    graph.update(None, None, use_inactive=True)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    for _ in range(6):
        graph.update(None, None, use_inactive=True)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=45, max_name_column_width=80))
