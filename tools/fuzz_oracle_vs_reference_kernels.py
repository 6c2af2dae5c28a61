"""Randomised comparison of oracle/droid_oracle.py with the reference's own kernels (oracle/_ref, CPU; see oracle/build_ref.py):
random windows of 4-9 keyframes with 1-3 edges per keyframe on the 12x16 maps -- RGB-D and monocular, motion-only every fifth,
a window start t0 > 1 every third, 0-2 px of target noise -- two Gauss-Newton iterations of `ba` each, plus `frame_distance`,
`projmap` (asserted bit-equal) and the fp16 lookup on pyramid sizes 12x16 / 15x20 / 7x10 / 3x5 (asserted bit-equal).
Runs for `seconds` (default 420) and prints the worst absolute differences.      python tools/fuzz_oracle_vs_reference_kernels.py [seconds]
Round-4 run: 108 problems -- poses 1.1e-6, dx 9.8e-7, disparities / dz 1.6e-5, frame_distance 1.2e-7."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import build_ref, droid_oracle as DO
from go_slam_amd import synth
R = build_ref.load()
worst = {}
def upd(k, a, b):
    e = float((a.double()-b.double()).abs().max()); worst[k] = max(worst.get(k, 0.0), e)
t0 = time.time(); n_ba = 0
for seed in range(100, 100000):
    if time.time() - t0 > (float(sys.argv[1]) if len(sys.argv) > 1 else 420.0):
        break
    g = torch.Generator().manual_seed(seed)
    nkf = int(torch.randint(4, 10, (1,), generator=g)); ne = int(torch.randint(nkf, 3*nkf, (1,), generator=g))
    rgbd = bool(seed % 2)
    p = synth.make_ba_problem(nkf, ne, "tiny", seed, rgbd)
    c, _ = DO.reproject(p["poses"], p["disps"], p["intrinsics"], p["ii"], p["jj"])
    p = synth.make_ba_problem(nkf, ne, "tiny", seed, rgbd, noise_px=float(torch.rand(1, generator=g))*2, coords=c[0])
    # random window start and a stereo edge now and then
    if seed % 3 == 0 and nkf > 5:
        t0w = int(torch.randint(1, nkf-2, (1,), generator=g)); p["t0"] = t0w
        kx = torch.unique(torch.cat([torch.arange(t0w, p["t1"]), p["ii"]]))
        ht, wd, _ = synth.SHAPES["tiny"]
        p["eta"] = 1e-2*torch.rand(len(kx), ht, wd, generator=g) + 1e-4
    K = p["intrinsics"][0].contiguous()
    mo = (seed % 5 == 0)
    outs=[]
    for fn in (R.ba, DO.ba):
        po, do = p["poses"].clone(), p["disps"].clone()
        res = fn(po, do, K, p["disps_sens"], p["target"], p["weight"], p["eta"], p["ii"], p["jj"], p["t0"], p["t1"], 2, 1e-4, 0.1, mo)
        outs.append((res, po, do))
    (rr, pr, dr), (ro, po, do) = outs
    upd("poses", po, pr); upd("disps", do, dr); upd("dx", ro[0], rr[0])
    if not mo: upd("dz", ro[1], rr[1])
    n_ba += 1
    # geometry on the same video
    ii, jj = p["ii"], p["jj"]
    upd("frame_distance", DO.frame_distance(p["poses"], p["disps"], K, ii, jj, 0.3), R.frame_distance(p["poses"], p["disps"], K, ii, jj, 0.3))
    assert torch.equal(DO.projmap(p["poses"], p["disps"], K, ii, jj)[0], R.projmap(p["poses"], p["disps"], K, ii, jj)[0]), seed
    # lookup on odd pyramid sizes
    h2, w2 = [(7,10),(3,5),(15,20),(12,16)][seed % 4]
    vol = torch.randn(2, 6, 9, h2, w2, generator=g).half()
    ys, xs = torch.meshgrid(torch.arange(6.), torch.arange(9.), indexing="ij")
    co = (torch.stack([xs*w2/9, ys*h2/6], 0)[None].repeat(2,1,1,1) + 3*torch.randn(2,2,6,9, generator=g)).contiguous()
    assert torch.equal(DO.corr_index_forward(vol, co, 3)[0], R.corr_index_forward(vol, co, 3)[0]), seed
print("ba problems", n_ba, "worst abs errors", {k: f"{v:.2e}" for k, v in worst.items()})
