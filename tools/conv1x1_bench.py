"""Timing of gs_conv1x1 on the update operator's two 1x1 layers (corr_encoder[0] 196 -> 128 on 75 edges, upmask 128 -> 576
on 25 keyframes, 60x80 maps): microseconds and achieved HBM GB/s (input rows + output rows, once each)."""
import json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import droid_net as DN
import bench
dev = torch.device("cuda:0")
out = {}
for name, n, k, o, act in (("corr_encoder0", 75, 196, 128, "relu"), ("upmask", 25, 128, 576, "none"), ("gru_w", 75, 128, 128, "none")):
    conv = torch.nn.Conv2d(k, o, 1).to(dev)
    x = torch.randn(n, k, 60, 80, device=dev).half().contiguous(memory_format=torch.channels_last)
    cache = {}
    ms = bench.time_op(lambda: DN.conv1x1_bias_act(cache, conv, x, act), iters=20, warm=3)
    y = DN.conv1x1_bias_act(cache, conv, x, act)
    ref = torch.nn.functional.conv2d(x.float(), conv.weight.half().float(), conv.bias.float())
    if act == "relu":
        ref = ref.relu()
    byts = n * 4800 * (k + o) * 2
    out[name] = {"us": round(ms * 1e3, 1), "GBps": round(byts / ms / 1e6, 1), "max_err": float((y.float() - ref).abs().max())}
print(json.dumps(out))
