"""Measured error of droid_backends.ba at the bench window (S480, P = 25, E = 75, 2 GN iterations) against the CPU
oracle: what tolerance `dx` / `dz` actually meet (SURVEY 8c asks rtol 1e-4 / atol 1e-6 on dx)."""
import json
import sys

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from go_slam_amd import droid_backends as db, synth     # noqa: E402
from oracle import droid_oracle as O                    # noqa: E402

dev = torch.device("cuda:0")
rep = {}
for shape, rgbd in (("S480", True), ("S480", False), ("Rep", True)):
    p = synth.make_ba_problem(25, 75, shape, 101, rgbd)
    c, _ = O.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
    p = synth.make_ba_problem(25, 75, shape, 101, rgbd, noise_px=0.5, coords=c[0])
    K = p["intrinsics"][0].contiguous()
    for iters in (1, 2):
        po, do = p["poses"].clone(), p["disps"].clone()
        ref = O.ba(po, do, K, p["disps_sens"], p["target"], p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"],
                   iters, 1e-4, 0.1, False)
        pg, dg = p["poses"].clone().to(dev), p["disps"].clone().to(dev)
        out = db.ba(pg, dg, K.to(dev), p["disps_sens"].to(dev), p["target"].to(dev), p["weight"].to(dev),
                    p["eta"].to(dev), p["ii"].to(dev), p["jj"].to(dev), p["t0"], p["t1"], iters, 1e-4, 0.1, False)
        torch.cuda.synchronize()
        dx, rdx = out[0].cpu().double(), ref[0].double()
        dz, rdz = out[1].cpu().double(), ref[1].double()
        e = (dx - rdx).abs()
        rep[f"{shape}_{'rgbd' if rgbd else 'mono'}_it{iters}"] = {
            "dx_absmax": float(rdx.abs().max()), "dx_err_max": float(e.max()),
            "dx_rel_l2": float(e.norm() / rdx.norm()),
            "dx_needed_rtol_at_atol1e-6": float(((e - 1e-6).clamp(min=0) / rdx.abs().clamp(min=1e-12)).max()),
            "dz_err_max": float((dz - rdz).abs().max()), "dz_absmax": float(rdz.abs().max()),
            "dz_rel_l2": float((dz - rdz).norm() / rdz.norm()),
            "pose_err_max": float((pg.cpu() - po).abs().max()), "disp_err_max": float((dg.cpu() - do).abs().max())}
print(json.dumps(rep, indent=1))
