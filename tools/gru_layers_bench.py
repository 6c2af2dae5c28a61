"""The update operator's convolution launches in their production form (fused epilogues), one by one, at the bench shape:
us per call with bare C-ABI launches.  
One JSON line."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import go_slam_amd.droid_net as DN  # noqa: E402
from go_slam_amd import _lib  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    n, h, w = 75, 60, 80
    cl = torch.channels_last
    gru = DN.ConvGRU(128, 320).to(dev).eval()
    L, st = _lib.lib(), _lib.stream_ptr(dev)
    net = (0.5 * torch.randn(n, 128, h, w, device=dev)).half().contiguous(memory_format=cl)
    hx = (0.5 * torch.randn(n, 320, h, w, device=dev)).half().contiguous(memory_format=cl)
    inp = (0.5 * torch.randn(n, 128, h, w, device=dev)).half().contiguous(memory_format=cl)
    wzr, wq, bzr, bq, ww, bw, gw = gru._half_weights()
    inp_pre = gru.inp_gates(inp)
    wzr, wq = gru._hw_hoist[1], gru._hw_hoist[2]
    gzr = torch.randn(n, 256, device=dev)
    gq = torch.randn(n, 128, device=dev)
    z = torch.empty_like(net); rnet = torch.empty_like(net); out = torch.empty_like(net)
    izr, iq = DN.conv3x3_weight_image(wzr, 32), DN.conv3x3_weight_image(wq, 32)
    res = {}
    res["gru_zr_us"] = 1e3 * bench.time_op(lambda: L.gs_conv3x3_gru_zr(_lib.ptr(hx), 320, 320, _lib.ptr(izr), _lib.ptr(bzr), _lib.ptr(gzr), _lib.ptr(inp_pre), _lib.ptr(z), _lib.ptr(rnet), n, h, w, st), iters=20, warm=5)
    res["gru_q_us"] = 1e3 * bench.time_op(lambda: L.gs_conv3x3_gru_q(_lib.ptr(rnet), hx.data_ptr() + 256, 320, 192, _lib.ptr(iq), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z), _lib.ptr(net), _lib.ptr(out), n, h, w, st), iters=20, warm=5)
    conv = torch.nn.Conv2d(128, 128, 3, padding=1).to(dev)
    cache = DN._HalfWeights()
    x = (0.5 * torch.randn(n, 128, h, w, device=dev)).half().contiguous(memory_format=cl)
    res["bias_relu_128_us"] = 1e3 * bench.time_op(lambda: DN.conv_bias_act(cache, conv, x, "relu", out=hx, out_channel=128), iters=20, warm=5)
    heads = torch.nn.Conv2d(128, 384, 3, padding=1).to(dev)
    wh = heads.weight.detach().half().contiguous(memory_format=cl)
    res["heads_us"] = 1e3 * bench.time_op(lambda: DN.conv3x3_hip(x, wh), iters=20, warm=5)
    print(json.dumps({k: round(v, 1) for k, v in res.items()}))


if __name__ == "__main__":
    main()
