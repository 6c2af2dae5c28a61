import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev=torch.device('cuda:0')
video, op, graph, _ = bench.build_state(dev)
coords1,_=video.reproject(graph.ii, graph.jj)
for _ in range(5):
    graph.corr(coords1)
torch.cuda.synchronize()
