"""gs_gru_glo_fused (w(net) + sigmoid + pooling + the three glo mat-vecs) at the bench shape: us per call, bare C-ABI
launches.  One JSON line."""
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
    gru = DN.ConvGRU(128, 320).to(dev).eval()
    out = {}
    for name, (n, h, w) in {"bench_75x60x80": (75, 60, 80), "scannet_78x30x40": (78, 30, 40)}.items():
        net = (0.7 * torch.randn(n, 128, h, w, device=dev)).half().contiguous(memory_format=torch.channels_last)
        wzr, wq, bzr, bq, ww, bw, gw = gru._half_weights()
        L, st = _lib.lib(), _lib.stream_ptr(dev)
        hw = h * w
        gzr = torch.empty(n, 256, dtype=torch.float32, device=dev)
        gq = torch.empty(n, 128, dtype=torch.float32, device=dev)
        ws = torch.empty(L.gs_gru_glo_fused_workspace_bytes(n, hw), dtype=torch.uint8, device=dev)
        args = (_lib.ptr(net), 128, _lib.ptr(gru._ww_pack), _lib.ptr(bw), _lib.ptr(gw[0]), _lib.ptr(gw[1]), _lib.ptr(gw[2]),
                _lib.ptr(gw[3]), _lib.ptr(gw[4]), _lib.ptr(gw[5]), _lib.ptr(gzr), _lib.ptr(gq), n, hw, _lib.ptr(ws),
                ws.numel(), st)
        us = 1e3 * bench.time_op(lambda: L.gs_gru_glo_fused(*args), iters=200, warm=20)
        out[name] = {"us": round(us, 1), "GBps_net_read": round(n * hw * 256 / us / 1e3, 0)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
