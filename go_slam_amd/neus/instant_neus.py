"""InstantNeuS with the reference's module tree and forward signature
(src/InstantNeuS.py:208-370), evaluated by the fused HIP pipeline `gs_neus_forward`:
hash-grid encode + SDF linear + analytic SDF gradient + NeuS alpha + MFMA colour MLP +
compositing in four launches, instead of ~60 ATen kernels, six boolean-index scatters and an
autograd graph with create_graph=True.

Parameter names match the reference so checkpoints interchange:
  sdf_network.encoding.encoding.params, sdf_network.encoding._B, sdf_network.sdf_layer.{weight,bias},
  color_network._B, color_network.network.params, variance_network.variance.
"""
import math

import torch
import torch.nn as nn

from .. import _lib
from ..droid_backends import _workspace
from .tcnn_compat import Encoding as TcnnEncoding, Network as TcnnNetwork, _require_inference


class Encoding(nn.Module):
    """src/InstantNeuS.py:35-94 (direction=False): tcnn HashGrid + the raw xyz."""

    def __init__(self, n_input_dims=3, device="cuda:0", direction=False):
        super().__init__()
        assert not direction, "the reference's colour net does not use the direction encoding"
        self.n_input_dims = n_input_dims
        self.include_xyz = True
        self.direction = direction
        self.encoding = TcnnEncoding(n_input_dims, dict(otype="HashGrid", n_levels=16, n_features_per_level=2,
                                                        log2_hashmap_size=19, base_resolution=16,
                                                        per_level_scale=1.447269237440378))
        self._B = nn.Parameter(torch.randn(n_input_dims, 3) * 25.0)
        self.n_output_dims = 3 + self.encoding.n_output_dims

    def forward(self, x):
        out = self.encoding((x + 1) / 2)
        return torch.cat([x, out.float()], dim=-1)


class SDFNetwork(nn.Module):
    """src/InstantNeuS.py:97-159."""

    def __init__(self, d_in=3, d_out=32, device="cuda:0"):
        super().__init__()
        self.d_in, self.d_out = d_in, d_out
        self.encoding = Encoding(n_input_dims=d_in, device=device)
        self.sdf_layer = nn.Linear(self.encoding.n_output_dims, d_out)
        nn.init.constant_(self.sdf_layer.bias, 0.0)
        nn.init.constant_(self.sdf_layer.weight[:, 3:], 0.0)
        nn.init.normal_(self.sdf_layer.weight[:, :3], mean=0.0, std=math.sqrt(2) / math.sqrt(d_out))

    def get_training_parameters(self, ignore_keys=()):
        return {"network": list(self.sdf_layer.parameters()) + [self.encoding._B],
                "volume": list(self.encoding.encoding.parameters())}


class ColorNetwork(nn.Module):
    """src/InstantNeuS.py:162-205."""

    def __init__(self, d_in=3, d_feat=31, d_hidden=64, n_layers=2, device="cuda:0"):
        super().__init__()
        self._B = nn.Parameter(torch.randn(3, 33) * 25.0)
        self.network = TcnnNetwork(33 + 3 + d_feat, 3, dict(otype="FullyFusedMLP", activation="ReLU",
                                                           output_activation="none", n_neurons=d_hidden,
                                                           n_hidden_layers=n_layers))


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val=0.2, scale_factor=10.0):
        super().__init__()
        self.scale_factor = scale_factor
        self.variance = nn.Parameter(torch.tensor(float(init_val)))

    def forward(self, x):
        return torch.ones(x.shape[0], 1, device=x.device) * torch.exp(self.variance * self.scale_factor)


class InstantNeuS(nn.Module):
    def __init__(self, cfg, bound, device="cuda:0"):
        super().__init__()
        self.cfg = cfg
        self.register_buffer("bound", torch.tensor(bound).float())
        self.register_buffer("realtime_bound", torch.tensor(bound).float())
        self.device = device
        self.sdf_network = SDFNetwork(**cfg.get("sdf_network", {}), device=device)
        self.color_network = ColorNetwork(**cfg.get("color_network", {}), device=device)
        self.variance_network = SingleVarianceNetwork(**cfg.get("variance_network", {}))
        self.sdf_smooth_std = cfg.get("sdf_smooth_std", 0.005)
        self.sdf_sparse_factor = cfg.get("sdf_sparse_factor", 5)
        self.sdf_truncation = cfg.get("sdf_truncation", 0.16)
        self.sdf_random_weight = cfg.get("sdf_random_weight", 0.04)
        self.cos_anneal_ratio = 1.0
        self._host_bounds = None

    def get_training_parameters(self, ignore_keys=()):
        groups = {"sdf_network": list(self.sdf_network.get_training_parameters()["network"]),
                  "color_network": list(self.color_network.parameters()),
                  "variance_network": list(self.variance_network.parameters())}
        return [p for k, v in groups.items() if k not in ignore_keys for p in v]

    def get_volume_parameters(self):
        return list(self.sdf_network.get_training_parameters()["volume"])

    @torch.no_grad()
    def update_bound(self, bound):
        self.realtime_bound[:] = bound.float().to(self.realtime_bound.device)
        self._host_bounds = None

    def _bounds_host(self):
        """(bound, realtime_bound) as two 6-float host arrays; cached so the hot path has no D2H."""
        if self._host_bounds is None:
            import ctypes
            b = self.bound.detach().cpu().reshape(-1).tolist()
            r = self.realtime_bound.detach().cpu().reshape(-1).tolist()
            self._host_bounds = ((ctypes.c_float * 6)(*b), (ctypes.c_float * 6)(*r))
        return self._host_bounds

    def forward(self, rays_o, rays_d, z_vals, dists, render_params=None):
        """src/InstantNeuS.py:295-370; returns the same dict of 9 tensors."""
        net = self.sdf_network
        _require_inference(rays_o, rays_d, net.encoding.encoding.params, net.sdf_layer.weight,
                           self.color_network.network.params)
        dev = rays_o.device
        n, s = z_vals.shape
        f32 = dict(dtype=torch.float32, device=dev)
        color, normal = torch.empty(n, 3, **f32), torch.empty(n, 3, **f32)
        depth, dvar, wsum, gerr = (torch.empty(n, 1, **f32) for _ in range(4))
        sdf, zmid = torch.empty(n, s, **f32), torch.empty(n, s, **f32)
        var = float(self.variance_network.variance.detach())          # host scalar (1 small D2H)
        inv_s = min(max(math.exp(var * self.variance_network.scale_factor), 1e-6), 1e6)
        bh, rh = self._bounds_host()
        L = _lib.lib()
        ws = _workspace(dev, L.gs_neus_forward_workspace_bytes(n, s) + 256)
        grid = net.encoding.encoding.params_half()
        mlp = self.color_network.network.params_half()
        args = [rays_o.detach().float().contiguous(), rays_d.detach().float().contiguous(),
                z_vals.detach().float().contiguous(), dists.detach().float().contiguous(), grid,
                net.sdf_layer.weight.detach().float().contiguous(), net.sdf_layer.bias.detach().float().contiguous(),
                self.color_network._B.detach().float().contiguous(), mlp]
        with torch.cuda.device(dev):
            rc = L.gs_neus_forward(*[_lib.ptr(a) for a in args], float(inv_s), bh, rh,
                                   _lib.ptr(color), _lib.ptr(depth), _lib.ptr(dvar), _lib.ptr(normal), _lib.ptr(wsum),
                                   _lib.ptr(sdf), _lib.ptr(zmid), _lib.ptr(gerr), None, None, None, n, s,
                                   _lib.ptr(ws), ws.numel(), _lib.stream_ptr(dev))
        _lib.check(rc, "InstantNeuS.forward")
        return {
            "color": color, "depth": depth, "depth_variance": dvar, "normal": normal, "weight_sum": wsum,
            "sdf_variance": torch.full((n, 1), 1.0 / math.exp(var * self.variance_network.scale_factor), **f32),
            "sdf": sdf, "z_vals": zmid, "gradient_error": (gerr.sum() / float(n * s)).unsqueeze(0),
        }

    def compute_sdf_error(self, sdf, z_vals, gt_depth):
        """src/InstantNeuS.py:372-400 (plain PyTorch, as in the reference)."""
        n, s = z_vals.shape
        pred = sdf.reshape(n, s)
        gt = gt_depth.reshape(n, 1)
        vm = (gt > 0).reshape(-1)
        gt, z, pred = gt[vm], z_vals[vm], pred[vm]
        front = z < (gt - self.sdf_truncation)
        bnd = gt - z
        sm = bnd.abs() <= self.sdf_truncation
        nvs = front.sum(1) + sm.sum(1) + 1e-8
        nvr = vm.sum()
        fl = torch.max(torch.exp((-self.sdf_sparse_factor * pred).clamp(max=10.0)) - torch.ones_like(pred),
                       pred - bnd).clamp(min=0.0) * front
        return ((torch.abs(pred - bnd) * sm).sum(1) / nvs).sum() / nvr, (fl.sum(1) / nvs).sum() / nvr
