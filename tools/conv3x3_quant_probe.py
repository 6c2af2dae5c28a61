"""How much does tile quantisation really cost?  conv3x3_pp (320 -> 256, 60x80) for edge counts that fill 5.0, 5.25, 5.5,
5.75 and 6.0 rounds of 256 workgroups: a quantised machine takes 6 rounds for everything above 5.0."""
import sys, os, json
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from go_slam_amd import droid_net as DN
dev = torch.device("cuda:0")
torch.manual_seed(0)
out = []
for c, o in ((320, 256), (320, 128), (128, 384)):
    w = (torch.randn(o, c, 3, 3, device=dev) / (3.0 * c ** 0.5)).half()
    for n in (64, 68, 71, 75, 78, 82, 85, 96, 102, 136):
        x = torch.randn(n, c, 60, 80, device=dev).half().contiguous(memory_format=torch.channels_last)
        ms = min(bench.time_op(lambda: DN.conv3x3_hip(x, w), iters=10, warm=3) for _ in range(3))
        tiles = -(-n * 60 // 32) * 5 * (o // 128)
        out.append({"c_in": c, "c_out": o, "edges": n, "workgroups": tiles, "rounds": tiles / 256.0, "us": ms * 1e3,
                    "us_per_round_of_work": ms * 1e3 / (tiles / 256.0)})
        print(out[-1])
json.dump(out, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "quant_probe.json"), "w"), indent=1)
