"""torch.profiler table of the dense-BA launches alone on the bench workload (P=25, E=75, 60x80)."""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device('cuda:0')
video, op, graph, _ = bench.build_state(dev)
for _ in range(3):
    graph.update(None, None, use_inactive=True)
idx = graph._edge_index(None, None, True)
ht, wd = graph.ht, graph.wd
target = torch.cat([graph.target_inac[:, idx["sel"]], graph.target], 1).view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
weight = torch.cat([graph.weight_inac[:, idx["sel"]], graph.weight], 1).view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
damping = 0.2 * graph.damping[idx["damping_index"]].contiguous() + 1e-7
def run():
    video.ba(target, weight, damping, idx["ii"], idx["jj"], t0=idx["t0"], t1=idx["t1"], iters=2, lm=1e-4, ep=0.1)
for _ in range(5):
    run()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CUDA]) as prof:
    for _ in range(20):
        run()
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=14, max_name_column_width=70))
