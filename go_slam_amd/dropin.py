"""One-call module substitution so the unmodified reference tree (`src/*.py`) resolves its native
imports to this package:

    import go_slam_amd.dropin as dropin; dropin.install()
    from src.slam import SLAM          # droid_backends / tinycudann / lietorch / torch_scatter resolve here
"""
import sys
import types

import torch


def _torch_scatter():
    m = types.ModuleType("torch_scatter")

    def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
        """torch_scatter.scatter_mean for a 1-D index along `dim` (the only form GO-SLAM uses,
        src/droid_net.py:59)."""
        dim = dim if dim >= 0 else src.dim() + dim
        n = int(dim_size) if dim_size is not None else int(index.max()) + 1
        shape = list(src.shape)
        shape[dim] = n
        acc = torch.zeros(shape, dtype=torch.float32, device=src.device)
        acc.index_add_(dim, index, src.float())
        cnt = torch.zeros(n, dtype=torch.float32, device=src.device)
        cnt.index_add_(0, index, torch.ones_like(index, dtype=torch.float32))
        view = [1] * src.dim()
        view[dim] = n
        return (acc / cnt.clamp(min=1).view(view)).to(src.dtype)
    m.scatter_mean = scatter_mean
    return m


def install(droid_backends=True, tinycudann=True, lietorch=True, torch_scatter=True):
    if droid_backends:
        from . import droid_backends as db
        sys.modules["droid_backends"] = db
    if tinycudann:
        from .neus import tcnn_compat
        sys.modules["tinycudann"] = tcnn_compat
    if lietorch:
        from . import lietorch_shim
        sys.modules["lietorch"] = lietorch_shim
    if torch_scatter and "torch_scatter" not in sys.modules:
        sys.modules["torch_scatter"] = _torch_scatter()
