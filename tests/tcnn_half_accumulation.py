#!/usr/bin/env python
"""How far is the oracle's reading of tiny-cuda-nn (fp32 accumulation of the 8 grid corners and of the MLP's dot products, one
rounding to fp16) from a half-accumulating one (upstream's published types: `vector_t<__half>` result in kernel_grid, half wmma
accumulator fragments in FullyFusedMLP)?  tiny-cuda-nn is absent from /root/reference and unpinned, so neither can be
run; this measures the distance between the two restatements on the trained-like grid, per stage and end to end through
`InstantNeuS.forward`'s restatement, against SURVEY's fp16-level tolerance (rtol 5e-3 / atol 1e-3).  CPU only.

    python tests/tcnn_half_accumulation.py            # -> profiles/r06_tcnn_half_accumulation.json
(lives under tests/: it executes the oracle, which only test infrastructure may do)
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _cmp(a, b):
    a, b = a.double(), b.double()
    d = (a - b).abs()
    return {"max_abs": float(d.max()), "rel_l2": float(d.norm() / b.norm().clamp(min=1e-30)), "ref_max_abs": float(b.abs().max()),
            "frac_outside_5e-3_1e-3": float((d > 1e-3 + 5e-3 * b.abs()).double().mean())}


def report(n_rays=300, seed=8, grid_init=0.3):
    from oracle import neus_oracle as NO
    P = NO.make_params(7, grid_init=grid_init, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
    P["rt_bound"] = torch.tensor([[-2.2, 2.3], [-2.4, 2.1], [-2.0, 2.2]])
    g = torch.Generator().manual_seed(seed)
    o = torch.rand(n_rays, 3, generator=g) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(n_rays, 3, generator=g), dim=1)
    gt = torch.rand(n_rays, generator=g) * 3.5 + 0.5
    z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, torch.rand(24, generator=g))
    x = torch.rand(20000, 3, generator=g)
    out = {"points": int(x.shape[0]), "rays": n_rays, "grid_init": grid_init,
           "what": "distance of the half-accumulating restatements FROM the fp32-accumulating one (the oracle default, what the "
                   "HIP kernels follow); SURVEY 8c tolerance for sdf / feat / colour: rtol 5e-3, atol 1e-3"}
    enc32 = NO.grid_encode(x, P["grid"], accumulate="float").float()
    for mode in ("half", "half_mul_add"):
        e = NO.grid_encode(x, P["grid"], accumulate=mode).float()
        out[f"grid_encode_{mode}"] = _cmp(e, enc32)
        out[f"grid_encode_{mode}"]["max_in_fp16_ulps_of_level_amplitude"] = float(
            ((e - enc32).abs().reshape(-1, 16, 2).amax(dim=(0, 2)) / (2.0 ** -10 * enc32.abs().reshape(-1, 16, 2).amax(dim=(0, 2)))).max())
    mi = torch.cat([torch.sin(x @ P["color_B"]), torch.nn.functional.normalize(torch.randn(x.shape[0], 3, generator=g), dim=1),
                    torch.randn(x.shape[0], 31, generator=g) * 0.3], 1)
    m32 = NO.mlp_forward(mi, P["mlp"], accumulate="float").float()
    out["mlp_forward_half"] = _cmp(NO.mlp_forward(mi, P["mlp"], accumulate="half").float(), m32)
    ref = NO.neus_forward(o, d, z, dist, P)
    for mode in ("half", "half_mul_add"):
        keep = NO.ACCUMULATE
        NO.ACCUMULATE = mode
        try:
            alt = NO.neus_forward(o, d, z, dist, P)
        finally:
            NO.ACCUMULATE = keep
        assert torch.equal(alt["_mask"], ref["_mask"])
        out[f"neus_forward_{mode}"] = {k: _cmp(alt[k].float(), ref[k].float())
                                       for k in ("sdf", "color", "depth", "normal", "weight_sum", "gradient_error", "_alpha", "_rgb")}
    return out


if __name__ == "__main__":
    r = report()
    path = os.path.join(ROOT, "profiles", "r06_tcnn_half_accumulation.json")
    json.dump(r, open(path, "w"), indent=1, sort_keys=True)
    print(json.dumps(r, indent=1, sort_keys=True))
