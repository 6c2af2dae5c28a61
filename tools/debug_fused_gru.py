"""Where do the fused ConvGRU epilogues differ from conv + gate kernels?  Prints per-tensor mismatch statistics."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_slam_amd.droid_net as DN
from go_slam_amd import _lib

dev = "cuda:0"
torch.manual_seed(21)
gru = DN.ConvGRU(128, 320).to(dev).eval()
n, h, w = 5, 32, 48
g = torch.Generator().manual_seed(22)
cl = lambda t: t.half().to(dev).contiguous(memory_format=torch.channels_last)
net = cl(torch.tanh(torch.randn(n, 128, h, w, generator=g)))
inp = cl(torch.relu(torch.randn(n, 128, h, w, generator=g)))
rest = cl(torch.relu(torch.randn(n, 192, h, w, generator=g)))
L = _lib.lib(); st = _lib.stream_ptr(torch.device(dev))
with torch.no_grad():
    wzr, wq, bzr, bq, ww, bw, gw = gru._half_weights()
    wzr, wq = gru._hw_hoist[1], gru._hw_hoist[2]
    inp_pre = gru.inp_gates(inp)
    hx = cl(torch.cat([net, rest], 1).float())
    hx0 = hx.clone()
    gzr = torch.randn(n, 256, device=dev) * 0.1
    gq = torch.randn(n, 128, device=dev) * 0.1
    cin = 320
    # unfused
    zr_pre = DN.conv3x3_hip(hx, wzr)
    z_u = torch.empty_like(net)
    _lib.check(L.gs_gru_gate_zr(_lib.ptr(zr_pre), _lib.ptr(bzr), _lib.ptr(gzr), _lib.ptr(inp_pre), _lib.ptr(hx), _lib.ptr(z_u), n, h * w, cin, st), "zr")
    rnet_u = hx[:, :128].clone()
    q_pre = DN.conv3x3_hip(hx, wq)
    out_u = torch.empty_like(net)
    _lib.check(L.gs_gru_gate_q(_lib.ptr(q_pre), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z_u), _lib.ptr(net), _lib.ptr(out_u), n, h * w, st), "q")
    # fused
    hx = hx0.clone()
    z_f = torch.empty_like(net); rnet_f = torch.empty_like(net); out_f = torch.empty_like(net)
    _lib.check(L.gs_conv3x3_gru_zr(_lib.ptr(hx), cin, cin, _lib.ptr(DN.conv3x3_weight_image(wzr, 32)), _lib.ptr(bzr), _lib.ptr(gzr), _lib.ptr(inp_pre), _lib.ptr(z_f), _lib.ptr(rnet_f), n, h, w, st), "fzr")
    _lib.check(L.gs_conv3x3_gru_q(_lib.ptr(rnet_f), hx.data_ptr() + 2 * 128, cin, cin - 128, _lib.ptr(DN.conv3x3_weight_image(wq, 32)), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z_f), _lib.ptr(net), _lib.ptr(out_f), n, h, w, st), "fq")
    # q path fed with the UNFUSED z / rnet, to isolate the second kernel
    out_f2 = torch.empty_like(net)
    _lib.check(L.gs_conv3x3_gru_q(_lib.ptr(rnet_u.contiguous(memory_format=torch.channels_last)), hx.data_ptr() + 2 * 128, cin, cin - 128, _lib.ptr(DN.conv3x3_weight_image(wq, 32)), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z_u), _lib.ptr(net), _lib.ptr(out_f2), n, h, w, st), "fq2")
    torch.cuda.synchronize()

def rep(name, a, b):
    a, b = a.float().permute(0, 2, 3, 1), b.float().permute(0, 2, 3, 1)      # n h w c
    d = (a - b).abs()
    bad = d > 0
    print(f"{name}: mismatches {int(bad.sum())}/{bad.numel()} max {float(d.max()):.3e}")
    if bad.any():
        idx = bad.nonzero()
        print("   first:", idx[:5].tolist(), " per-channel counts (first 16 nonzero):", [(int(c), int(k)) for c, k in enumerate(bad.sum((0, 1, 2)).tolist()) if k][:16])
        print("   rows with mismatches:", sorted(set(idx[:, 1].tolist()))[:20], " cols:", sorted(set(idx[:, 2].tolist()))[:20], " imgs:", sorted(set(idx[:, 0].tolist())))
rep("z", z_f, z_u); rep("r*net", rnet_f, rnet_u); rep("out (fused chain)", out_f, out_u); rep("out (q kernel on unfused z, rnet)", out_f2, out_u)
