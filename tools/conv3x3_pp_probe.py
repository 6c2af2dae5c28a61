"""Where does a conv3x3_pp workgroup spend its time?  s_memtime stamps (start / prologue done / main loop done / end) of
every workgroup + A/B of the variant bits, on the update operator's layer shapes.  Prints JSON lines.
Needs the probe build of the library: `make -C go_slam_amd/csrc clean all PROBES=1` (the shipped library has no probe
instantiation and no probe symbol)."""
import ctypes, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import _lib, droid_net as DN

LAYERS = {"gru_zr": (320, 256), "heads": (128, 384), "corr_enc2": (128, 128)}
dev = torch.device("cuda:0")
L = _lib.lib()
if not hasattr(L, "gs_conv3x3_pp_probe"):
    raise SystemExit("libgoslam_hip.so was built without PROBES=1")
_P = ctypes.c_void_p
L.gs_conv3x3_pp_probe.restype = ctypes.c_int
L.gs_conv3x3_pp_probe.argtypes = [_P, ctypes.c_int, ctypes.c_int, _P, ctypes.c_int, _P] + [ctypes.c_int] * 6 + [_P, _P]


def run(c, o, variant, dbg, x, wp, y, n, h, w):
    rc = L.gs_conv3x3_pp_probe(_lib.ptr(x), c, c, _lib.ptr(wp), 16, _lib.ptr(y), o, o, n, h, w, variant, _lib.ptr(dbg),
                               _lib.stream_ptr(dev))
    _lib.check(rc, "probe")


for name, (c, o) in LAYERS.items():
    n, h, w = 75, 60, 80
    x = torch.randn(n, h, w, c, device=dev).half()
    wt = (torch.randn(o, c, 3, 3, device=dev) / (3 * c ** 0.5)).half()
    wp = DN.pack_conv3x3_weight(wt, 32)
    y = torch.empty(n, h, w, o, device=dev, dtype=torch.float16)
    nwg = ((w + 15) // 16) * ((n * h + 31) // 32) * (o // 128)
    dbg = torch.zeros(nwg, 4, dtype=torch.int64, device=dev)
    for _ in range(3):
        run(c, o, 0, dbg, x, wp, y, n, h, w)
    torch.cuda.synchronize()
    d = dbg.cpu().double()
    span = float(d[:, 3].max() - d[:, 0].min())
    pro, loop, epi = (d[:, 1] - d[:, 0]), (d[:, 2] - d[:, 1]), (d[:, 3] - d[:, 2])
    out = {"layer": name, "workgroups": nwg, "ticks_kernel_span": span,
           "prologue_ticks_mean": float(pro.mean()), "loop_ticks_mean": float(loop.mean()), "epilogue_ticks_mean": float(epi.mean()),
           "loop_ticks_per_phase": float(loop.mean()) / (2 * 9 * (c // 32)),
           "wg_ticks_mean": float((d[:, 3] - d[:, 0]).mean()), "wg_ticks_sum_over_span_x256": float((d[:, 3] - d[:, 0]).sum()) / span / 256}
    times = {}
    for variant in (0, 1, 2, 4, 8, 12, 0, 1, 2, 4, 8, 12):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        for _ in range(2):
            run(c, o, variant, None, x, wp, y, n, h, w)
        s.record()
        for _ in range(10):
            run(c, o, variant, None, x, wp, y, n, h, w)
        e.record(); torch.cuda.synchronize()
        times.setdefault(variant, []).append(round(s.elapsed_time(e) / 10 * 1e3, 1))
    out["us_by_variant(0 base,1 noprio,2 nomask,4 no DMA in loop,8 no fragment reads in loop,12 neither)"] = times
    print(json.dumps(out))
