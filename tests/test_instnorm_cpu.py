"""CPU checks of the frame encoders' normalisation tail (go_slam_amd/csrc/instnorm.hip) through its NumPy restatement
tools/emulate_instnorm.py: the chunked shifted-sum / Chan-merge statistics against float64, and the rounding chain
against torch's op sequence on fp16 tensors.  On hardware: tests/test_widen_gpu.py::test_norm_act_*."""
import importlib.util
import os

import numpy as np
import torch


def _emu():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "emulate_instnorm.py")
    spec = importlib.util.spec_from_file_location("emulate_instnorm", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_chunked_statistics_match_float64():
    E = _emu()
    rng = np.random.default_rng(3)
    for hw, c, loc, scale in ((700, 32, 0.4, 1.7), (300, 64, -3.0, 0.05), (1000, 128, 10.0, 2.0), (257, 32, 0.0, 1e-3)):
        x = (rng.standard_normal((hw, c)) * scale + loc).astype(np.float16)
        mean, invstd = E.image_stats(x)
        x64 = x.astype(np.float64)
        ref_mean = x64.mean(0)
        ref_inv = 1.0 / np.sqrt(x64.var(0) + 1e-5)
        assert np.allclose(mean, ref_mean, rtol=2e-6, atol=2e-6 * max(1.0, abs(loc)))
        assert np.allclose(invstd, ref_inv, rtol=2e-5), (hw, c)       # (a large mean must not cancel the variance away)


def test_rounding_chain_matches_torch_ops_on_half_tensors():
    """relu(skip + relu(instance_norm(x + b))) with every intermediate an fp16 tensor, as ResidualBlock.forward runs it under
    autocast (src/modules/extractor.py:49-57)"""
    import torch.nn.functional as F
    E = _emu()
    g = torch.Generator().manual_seed(11)
    h, w, c = 12, 20, 32
    x = (torch.randn(1, c, h, w, generator=g) * 1.3 + 0.2).half()
    skip = torch.randn(1, c, h, w, generator=g).half()
    bias = (torch.randn(c, generator=g) * 0.5).half()
    xb = (x.float() + bias.float().view(1, c, 1, 1)).half()
    ref = F.relu((skip.float() + F.relu(F.instance_norm(xb.float()).half()).float()).half())
    to_rows = lambda t: t[0].permute(1, 2, 0).reshape(h * w, c).numpy()
    out = E.norm_act(to_rows(x), bias.numpy(), to_rows(skip), True, True, True)
    diff = np.abs(out.astype(np.float32) - to_rows(ref).astype(np.float32))
    tol = 2.0 ** -9 * (np.abs(to_rows(skip).astype(np.float32)) + np.abs(out.astype(np.float32))) + 1e-6
    assert (diff <= tol).all() and (diff > 0).mean() < 0.02
    plain = E.norm_act(to_rows(x), None, to_rows(skip), False, True, True)
    assert np.array_equal(plain, to_rows(F.relu((skip.float() + F.relu(x).float()).half())))
