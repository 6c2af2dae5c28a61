"""One-off evidence (CPU only, minutes): oracle/droid_oracle.py against the REFERENCE's own kernels (oracle/_ref, built by
oracle/build_ref.py) at the BENCH shapes -- the S480 frontend window (60x80 maps, P = 25, E = 75, RGB-D) and the monocular
window (40x80, P = 50, E = 100, no depth prior): two Gauss-Newton iterations of `ba`, `frame_distance`, `projmap`, and the
corr lookup on an S480 volume.  Prints one JSON object (-> profiles/r04_reference_kernels_parity.json)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from go_slam_amd import synth  # noqa: E402
from oracle import build_ref, droid_oracle as DO  # noqa: E402


def err(a, b):
    a, b = a.double(), b.double()
    return {"max_abs": float((a - b).abs().max()), "rel_l2": float((a - b).norm() / b.norm().clamp(min=1e-300)),
            "ref_max_abs": float(b.abs().max()), "bit_equal": bool(torch.equal(a, b))}


def ba_case(R, shape, num_kf, num_edges, rgbd, seed):
    p = synth.make_ba_problem(num_kf, num_edges, shape, seed, rgbd)
    c, _ = DO.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
    p = synth.make_ba_problem(num_kf, num_edges, shape, seed, rgbd, noise_px=0.5, coords=c[0])
    K = p["intrinsics"][0].contiguous()
    res = {}
    outs = []
    for name, fn in (("reference_kernels", R.ba), ("oracle", DO.ba)):
        po, do = p["poses"].clone(), p["disps"].clone()
        t = time.time()
        out = fn(po, do, K, p["disps_sens"], p["target"], p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], 2,
                 1e-4, 0.1, False)
        res[name + "_seconds"] = round(time.time() - t, 1)
        outs.append((out, po, do))
    (ro, pr, dr), (oo, po, do) = outs
    res.update(dx=err(oo[0], ro[0]), dz=err(oo[1], ro[1]), poses=err(po, pr), disps=err(do, dr),
               pose_step_max=float((pr - p["poses"]).abs().max()), unknowns=6 * (p["t1"] - p["t0"]))
    return res


def main():
    build_ref.build()
    R = build_ref.load()
    assert R is not None, "oracle/_ref is not built"
    out = {"what": "oracle/droid_oracle.py (first operand) vs the reference's own CUDA kernels compiled for the CPU "
                   "(oracle/_ref, second operand) at the bench shapes; errors of the oracle relative to the reference"}
    out["ba_S480_P25_E75_rgbd"] = ba_case(R, "S480", 25, 75, True, 31)
    print(json.dumps(out), flush=True)
    out["ba_Rep_P50_E100_mono"] = ba_case(R, "Rep", 50, 100, False, 33)
    vid = synth.make_video(25, "S480", seed=35)
    ii, jj = synth.make_graph(25, 75, seed=35)
    P, D, K = vid["poses"], vid["disps"], vid["intrinsics"][0].contiguous()
    out["frame_distance_S480_E75"] = err(DO.frame_distance(P, D, K, ii, jj, 0.3), R.frame_distance(P, D, K, ii, jj, 0.3))
    oc, ov = DO.projmap(P, D, K, ii, jj)
    rc, rv = R.projmap(P, D, K, ii, jj)
    out["projmap_S480_E75"] = {"coords": err(oc, rc), "valid": err(ov, rv)}
    ht, wd, _ = synth.SHAPES["S480"]
    g = torch.Generator().manual_seed(37)
    vol = torch.randn(2, ht, wd, ht, wd, generator=g).half()
    ys, xs = torch.meshgrid(torch.arange(ht, dtype=torch.float32), torch.arange(wd, dtype=torch.float32), indexing="ij")
    co = (torch.stack([xs, ys], 0)[None].repeat(2, 1, 1, 1) + 6.0 * torch.randn(2, 2, ht, wd, generator=g)).contiguous()
    out["corr_index_forward_S480_fp16"] = err(DO.corr_index_forward(vol, co, 3)[0], R.corr_index_forward(vol, co, 3)[0])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
