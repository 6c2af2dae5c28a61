"""`tinycudann`-compatible modules as far as the reference uses them
(src/InstantNeuS.py:62,66,77,86,116,192,201): `Encoding(n_input_dims, encoding_config)` with
`.n_output_dims` and one flat fp32 `params`; `Network(n_input_dims, n_output_dims,
network_config)` with `.params`.  Forward runs the HIP kernels behind include/goslam_neus.h.

state_dict keys stay `...encoding.params` / `...network.params` (checkpoints).
Round-1 status: inference (no autograd).  Calling with gradients enabled on a parameter that
requires grad raises instead of silently detaching.
"""
import ctypes
import math

import torch
import torch.nn as nn

from .. import _lib
from ..droid_backends import _workspace


def _require_inference(*tensors):
    if torch.is_grad_enabled() and any(t.requires_grad for t in tensors):
        raise NotImplementedError("go_slam_amd.neus: the backward (training) kernels are not built yet; "
                                  "call under torch.no_grad()")


class Encoding(nn.Module):
    """tcnn.Encoding for otype HashGrid with the InstantNeuS configuration (InstantNeuS.py:44-52)."""

    def __init__(self, n_input_dims=3, encoding_config=None, seed=1337, dtype=torch.float16):
        super().__init__()
        cfg = dict(encoding_config or {})
        if cfg.get("otype", "HashGrid") not in ("HashGrid", "Grid"):
            raise NotImplementedError(f"encoding otype {cfg.get('otype')} is not on the hot path")
        want = dict(n_levels=16, n_features_per_level=2, log2_hashmap_size=19, base_resolution=16)
        for k, v in want.items():
            if cfg.get(k, v) != v:
                raise NotImplementedError(f"HashGrid {k}={cfg[k]}: only the InstantNeuS configuration is built")
        if abs(cfg.get("per_level_scale", 1.447269237440378) - 1.447269237440378) > 1e-9:
            raise NotImplementedError("HashGrid per_level_scale must be 1.447269237440378")
        assert n_input_dims == 3
        self.n_input_dims = 3
        self.n_output_dims = 32
        self.dtype = dtype
        m = _lib.grid_meta() if _have_lib() else None
        total = int(m.total) if m is not None else 6299960
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid parameters U(-1e-4, 1e-4)
        self.params = nn.Parameter((torch.rand(total * 2, generator=g) * 2 - 1) * 1e-4)

    def params_half(self):
        return self.params.detach().to(torch.float16).contiguous()

    def forward(self, x, return_dy_dx=False):
        _require_inference(x, self.params)
        assert x.is_cuda and x.shape[-1] == 3
        x = x.detach().float().contiguous()
        n = x.shape[0]
        out = torch.empty(n, 32, dtype=torch.float16, device=x.device)
        dy = torch.empty(n, 32, 3, dtype=torch.float32, device=x.device) if return_dy_dx else None
        with torch.cuda.device(x.device):
            rc = _lib.lib().gs_grid_encode(_lib.ptr(x), _lib.ptr(self.params_half()), _lib.ptr(out), _lib.ptr(dy), n,
                                           _lib.stream_ptr(x.device))
        _lib.check(rc, "tcnn.Encoding")
        return (out, dy) if return_dy_dx else out


class Network(nn.Module):
    """tcnn.Network with otype FullyFusedMLP (InstantNeuS.py:184-192): width 64, 2 hidden layers,
    ReLU, no bias, input padded to a multiple of 16 with ones, output padded to 16."""

    def __init__(self, n_input_dims, n_output_dims, network_config=None, seed=1337):
        super().__init__()
        cfg = dict(network_config or {})
        if cfg.get("n_neurons", 64) != 64 or cfg.get("n_hidden_layers", 2) != 2 or \
                cfg.get("activation", "ReLU") != "ReLU" or cfg.get("output_activation", "none").lower() != "none":
            raise NotImplementedError("only FullyFusedMLP 64x2 ReLU/none (the InstantNeuS colour net) is built")
        if n_input_dims > 80 or n_output_dims > 4:
            raise NotImplementedError("FullyFusedMLP: <= 80 inputs and <= 4 outputs are built")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        pad_in = 80
        g = torch.Generator().manual_seed(seed)

        def xavier(o, i):
            a = math.sqrt(6.0 / (o + i))
            return ((torch.rand(o, i, generator=g) * 2 - 1) * a).reshape(-1)
        self.params = nn.Parameter(torch.cat([xavier(64, pad_in), xavier(64, 64), xavier(16, 64)]))

    def params_half(self):
        return self.params.detach().to(torch.float16).contiguous()

    def forward(self, x):
        _require_inference(x, self.params)
        assert x.is_cuda and x.shape[-1] == self.n_input_dims
        x = x.detach().to(torch.float16).contiguous()
        n = x.shape[0]
        out = torch.empty(n, self.n_output_dims, dtype=torch.float16, device=x.device)
        L = _lib.lib()
        need = L.gs_mlp_workspace_bytes(n, self.n_input_dims)
        ws = _workspace(x.device, need + 256)
        with torch.cuda.device(x.device):
            rc = L.gs_mlp_forward(_lib.ptr(x), _lib.ptr(self.params_half()), _lib.ptr(out), n, self.n_input_dims,
                                  self.n_output_dims, _lib.ptr(ws), ws.numel(), _lib.stream_ptr(x.device))
        _lib.check(rc, "tcnn.Network")
        return out


def _have_lib():
    import os
    return os.path.exists(_lib.LIB_PATH)
