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

def rep(name, a, b):
    a, b = a.float().permute(0, 2, 3, 1), b.float().permute(0, 2, 3, 1)      # n h w c
    d = (a - b).abs()
    bad = d > 0
    print(f"  {name}: mismatches {int(bad.sum())}/{bad.numel()} max {float(d.max()):.3e}")
    if bad.any():
        idx = bad.nonzero()
        print("     first:", idx[:4].tolist(), " channels:", sorted(set(idx[:, 3].tolist()))[:24])
        print("     rows:", sorted(set(idx[:, 1].tolist()))[:24], " cols:", sorted(set(idx[:, 2].tolist()))[:24], " imgs:", sorted(set(idx[:, 0].tolist())))

for hoisted in (True, False):
    print("hoisted", hoisted)
    with torch.no_grad():
        wzr, wq, bzr, bq, ww, bw, gw = gru._half_weights()
        if hoisted:
            wzr, wq = gru._hw_hoist[1], gru._hw_hoist[2]
            inp_pre = gru.inp_gates(inp)
            hx = cl(torch.cat([net, rest], 1).float())
        else:
            inp_pre = None
            hx = cl(torch.cat([net, inp, rest], 1).float())
        cin = hx.shape[1]
        hx0 = hx.clone()
        # the real global-context terms, as forward_hx computes them
        w_pre = torch.nn.functional.conv2d(net, ww, None)
        gzr = torch.empty(n, 256, dtype=torch.float32, device=dev); gq = torch.empty(n, 128, dtype=torch.float32, device=dev)
        ws = torch.empty(L.gs_gru_glo_workspace_bytes(n), dtype=torch.uint8, device=dev)
        _lib.check(L.gs_gru_glo(_lib.ptr(w_pre), _lib.ptr(bw), _lib.ptr(net), _lib.ptr(gw[0]), _lib.ptr(gw[1]), _lib.ptr(gw[2]), _lib.ptr(gw[3]), _lib.ptr(gw[4]), _lib.ptr(gw[5]), _lib.ptr(gzr), _lib.ptr(gq), n, h * w, _lib.ptr(ws), ws.numel(), st), "glo")
        zr_pre = DN.conv3x3_hip(hx, wzr)
        z_u = torch.empty_like(net)
        _lib.check(L.gs_gru_gate_zr(_lib.ptr(zr_pre), _lib.ptr(bzr), _lib.ptr(gzr), _lib.ptr(inp_pre), _lib.ptr(hx), _lib.ptr(z_u), n, h * w, cin, st), "zr")
        rnet_u = hx[:, :128].clone()
        q_pre = DN.conv3x3_hip(hx, wq)
        out_u = torch.empty_like(net)
        _lib.check(L.gs_gru_gate_q(_lib.ptr(q_pre), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z_u), _lib.ptr(net), _lib.ptr(out_u), n, h * w, st), "q")
        hx = hx0.clone()
        z_f = torch.empty_like(net); rnet_f = torch.empty_like(net); out_f = torch.empty_like(net); out_f2 = torch.empty_like(net)
        _lib.check(L.gs_conv3x3_gru_zr(_lib.ptr(hx), cin, cin, _lib.ptr(DN.conv3x3_weight_image(wzr, 32)), _lib.ptr(bzr), _lib.ptr(gzr), _lib.ptr(inp_pre), _lib.ptr(z_f), _lib.ptr(rnet_f), n, h, w, st), "fzr")
        _lib.check(L.gs_conv3x3_gru_q(_lib.ptr(rnet_f), hx.data_ptr() + 2 * 128, cin, cin - 128, _lib.ptr(DN.conv3x3_weight_image(wq, 32)), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z_f), _lib.ptr(net), _lib.ptr(out_f), n, h, w, st), "fq")
        _lib.check(L.gs_conv3x3_gru_q(_lib.ptr(rnet_u), hx.data_ptr() + 2 * 128, cin, cin - 128, _lib.ptr(DN.conv3x3_weight_image(wq, 32)), _lib.ptr(bq), _lib.ptr(gq), _lib.ptr(inp_pre), _lib.ptr(z_u), _lib.ptr(net), _lib.ptr(out_f2), n, h, w, st), "fq2")
        # and the module's own two paths
        outs = {}
        for fused in (False, True):
            DN.GRU_FUSED_EPILOGUE = fused
            outs[fused] = gru.forward_hx(net.clone(), hx0.clone(), inp_pre)
        torch.cuda.synchronize()
    print("  rnet_u strides", rnet_u.stride(), "net strides", net.stride())
    rep("z", z_f, z_u); rep("r*net", rnet_f, rnet_u); rep("out (fused chain)", out_f, out_u)
    rep("out (q kernel on unfused z, rnet)", out_f2, out_u)
    rep("forward_hx fused vs unfused", outs[True], outs[False]); rep("forward_hx unfused vs manual unfused", outs[False], out_u)
    rep("forward_hx fused vs manual fused", outs[True], out_f)
