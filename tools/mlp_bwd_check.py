"""gs_mlp_backward against a torch restatement with the same fp16 rounding points, and its launch time.

    python tools/mlp_bwd_check.py [--points 2359296 294912] [--out profiles/r06_mlp_bwd.json]

Run on the GPU box.  GS_MLP_BWD_NSUB=1|2 in the environment forces the kernel's points-per-iteration (A/B runs).
The comparison is a development aid (the parity tests are tests/test_neus_gpu.py: autograd on the oracle's network and
the reference module's own gradients); the timing is what profiles/ records.
"""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from go_slam_amd import _lib  # noqa: E402
if os.environ.get("GS_CHECK_LIB"):                 # A/B against another build of the library (same ABI)
    _lib.LIB_PATH = os.environ["GS_CHECK_LIB"]
from go_slam_amd.neus.tcnn_compat import _pack_mlp_fragments  # noqa: E402

LS = 128.0


def reference(X, W, d_rgb, rgb):
    W1, W2, W3 = W[:5120].view(64, 80).float(), W[5120:9216].view(64, 64).float(), W[9216:].view(16, 64).float()
    Xf = X.float()
    H1 = torch.relu(Xf @ W1.t()).half().float()
    H2 = torch.relu(H1 @ W2.t()).half().float()
    y = rgb.float()
    dpre = torch.zeros(X.shape[0], 16, device=X.device)
    dpre[:, :3] = (d_rgb * (y * (1 - y)) * LS).half().float()
    dH2 = ((dpre @ W3) * (H2 > 0)).half().float()
    dH1 = ((dH2 @ W2) * (H1 > 0)).half().float()
    dX = (dH1 @ W1).half()
    g = torch.cat([(dH1.t() @ Xf).reshape(-1), (dH2.t() @ H1).reshape(-1), (dpre.t() @ H2).reshape(-1)])
    return dX, g


def run(n, dev, check=True, iters=20):
    L = _lib.lib()
    g = torch.Generator(device="cpu").manual_seed(5)
    X = (torch.randn(n, 80, generator=g) * 0.5).half().to(dev)
    X[:, 67:] = 1.0
    W = (torch.randn(10240, generator=g) * 0.15).half().to(dev)
    d_rgb = (torch.randn(n, 3, generator=g) * 1e-3).to(dev)
    rgb = torch.rand(n, 3, generator=g).half().to(dev)
    wpack = _pack_mlp_fragments(W)
    nb = L.gs_mlp_backward_blocks(n)
    partial = torch.empty(nb, 10240, device=dev)
    dX = torch.empty(n, 80, dtype=torch.float16, device=dev)
    st = _lib.stream_ptr(dev)

    def launch():
        rc = L.gs_mlp_backward(_lib.ptr(X), _lib.ptr(wpack), _lib.ptr(d_rgb), _lib.ptr(rgb), LS, _lib.ptr(dX),
                               _lib.ptr(partial), n, st)
        _lib.check(rc, "mlp_backward")

    launch()
    torch.cuda.synchronize()
    res = {"points": n, "workgroups": nb}
    if check:
        m = min(n, 65536 + 37)                     # (the torch restatement on a prefix with a ragged end)
        Xs, ds, rs = X[:m].contiguous(), d_rgb[:m].contiguous(), rgb[:m].contiguous()
        nb2 = L.gs_mlp_backward_blocks(m)
        p2 = torch.empty(nb2, 10240, device=dev)
        dX2 = torch.empty(m, 80, dtype=torch.float16, device=dev)
        rc = L.gs_mlp_backward(_lib.ptr(Xs), _lib.ptr(wpack), _lib.ptr(ds), _lib.ptr(rs), LS, _lib.ptr(dX2), _lib.ptr(p2), m, st)
        _lib.check(rc, "mlp_backward")
        torch.cuda.synchronize()
        rdX, rg = reference(Xs, W, ds, rs)
        gk = p2.sum(0)

        def rel(a, b):
            return float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
        res.update(check_points=m, dX_rel_l2=rel(dX2, rdX), dW1_rel_l2=rel(gk[:5120], rg[:5120]),
                   dW2_rel_l2=rel(gk[5120:9216], rg[5120:9216]), dW3_rel_l2=rel(gk[9216:], rg[9216:]),
                   dX_max_abs=float((dX2.float() - rdX.float()).abs().max()), dX_ref_max=float(rdX.float().abs().max()))
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    res["us_per_launch"] = e0.elapsed_time(e1) * 1e3 / iters
    if os.environ.get("GS_MLP_BWD_STAMPS"):         # a -DMLPB_STAMP build leaves workgroup 0 / wave 0's phase times (100 MHz ticks)
        res["phase_us_wave0"] = [round(float(v) / 100.0, 2) for v in partial[0, :10].tolist()]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--points", type=int, nargs="*", default=[2359296, 294912])
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    out = {"forced_nsub": os.environ.get("GS_MLP_BWD_NSUB"), "runs": [run(n, dev) for n in a.points]}
    print(json.dumps(out))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
