"""`tinycudann`-compatible modules as far as the reference uses them
(src/InstantNeuS.py:62,66,77,86,116,192,201): `Encoding(n_input_dims, encoding_config)` with
`.n_output_dims` and one flat fp32 `params`; `Network(n_input_dims, n_output_dims,
network_config)` with `.params`.  Both are differentiable the way the reference needs them:

* `Encoding.__call__(x)` -> fp16 [N,32], differentiable w.r.t. `params` and `x`, and the input gradient is
  itself differentiable (tcnn's `_module_function_backward`): the reference calls
  `torch.autograd.grad(sdf, pts, create_graph=True)` inside `torch.enable_grad()` on EVERY forward
  (src/InstantNeuS.py:134-148) and later back-propagates the loss through that gradient (eikonal term, normals,
  the colour network's `normals` input).  First order = `gs_grid_backward(v=NULL)`, second order =
  `gs_grid_backward(v = dL/d(dx))`: d L / d dy (which reaches `sdf_layer.weight`), d L / d params and the mixed
  second derivatives w.r.t. x.
* `Network.__call__(x)` -> fp16 [N,n_out] (raw outputs, `output_activation: none`), differentiable w.r.t.
  `params` and `x` (first order, as tcnn's FullyFusedMLP): the one-launch MFMA backward `gs_mlp_backward`.

state_dict keys stay `...encoding.params` / `...network.params` (checkpoints); the modules
survive `share_memory()`, `copy.deepcopy`, `.to(device)` and `state_dict()` round trips (plain nn.Modules with
one Parameter each; the fp16 working copies are caches keyed on the parameter's version).
Kernels behind include/goslam_neus.h; there is no CPU path.
"""
import math
import os

import torch
import torch.nn as nn

from .. import _lib
from ..droid_backends import _workspace

GS_F16, GS_F32 = 0, 1
LOSS_SCALE = 128.0          # tcnn's default loss scale for fp16 gradients


def _have_lib():
    return os.path.exists(_lib.LIB_PATH)


class _HalfCache:
    """fp16 working copy of an fp32 master parameter, re-cast only when the parameter changes (tcnn keeps a
    persistent fp16 copy; re-casting the 12.6 M-entry grid on every call costs more than the encode itself).
    "Changes" = its storage address or autograd version counter: every in-place torch op, load_state_dict and the
    foreach optimisers advance the counter; `torch.optim.*(fused=True)` does NOT -- an optimiser of that kind must be
    followed by `torch.autograd.graph.increment_version(p)` (go_slam_amd.neus.mapper.make_optimizer registers a step
    hook that does it), or by `invalidate()`."""

    _key = None
    _val = None
    _pending = None     # callable or None: completes an exchange that is still writing `_val` (FlatAdamW's deferred
                        # all-gather of the updated fp16 table); every reader of the copy passes through get()

    def get(self, p):
        if self._pending is not None:
            pend, self._pending = self._pending, None
            pend()
        key = (p.data_ptr(), p._version, p.device, p.dtype)
        if self._key != key:
            self._val = p.detach().to(torch.float16).contiguous()
            self._key = key
        return self._val

    def invalidate(self):
        self._key = None
        self._val = None

    def __deepcopy__(self, memo):        # a copied module re-derives its cache
        return _HalfCache()

    def __reduce__(self):                # and so does a pickled one (mp spawn)
        return (_HalfCache, ())


def _dy_arg(dy):
    if dy.dtype == torch.float16:
        return dy.contiguous(), GS_F16
    return dy.float().contiguous(), GS_F32


def _grid_backward(x32, p16, dy, v, want_params, want_dx, want_ddy, grad_dtype):
    """One launch of gs_grid_backward; returns (dparams fp32 [total*2] | None, dx f32 [n,3] | None, ddy f32 | None)."""
    n = x32.shape[0]
    dev = x32.device
    dyc, dyt = _dy_arg(dy)
    gg = dx = ddy = None
    scale = 1.0
    if want_params:
        if grad_dtype == torch.float16:
            gg = torch.zeros(p16.numel(), dtype=torch.float16, device=dev)
            scale = LOSS_SCALE
        else:
            gg = torch.zeros(p16.numel(), dtype=torch.float32, device=dev)
    if want_dx:
        dx = torch.empty(n, 3, dtype=torch.float32, device=dev)
    if want_ddy:
        ddy = torch.empty(n, 32, dtype=torch.float32, device=dev)
    if n > 0 and (want_params or want_dx or want_ddy):
        with torch.cuda.device(dev):
            rc = _lib.lib().gs_grid_backward(
                _lib.ptr(x32), _lib.ptr(p16), _lib.ptr(dyc), dyt, 1.0, _lib.ptr(v), _lib.ptr(gg),
                GS_F16 if grad_dtype == torch.float16 else GS_F32, scale, _lib.ptr(dx), _lib.ptr(ddy), n,
                _lib.stream_ptr(dev))
        _lib.check(rc, "tcnn.Encoding backward")
    if gg is not None and gg.dtype == torch.float16:
        gg = gg.float().div_(LOSS_SCALE)
    return gg, dx, ddy


class _GridBackwardInput(torch.autograd.Function):
    """dx = (d y / d x)^T dy as a differentiable function of (dy, x, params) -- tcnn's
    `_module_function_backward`; its backward is tcnn's `bwd_bwd_input`."""

    @staticmethod
    def forward(ctx, dy, x, params, p16, grad_dtype):
        x32 = x.detach().float().contiguous()
        ctx.save_for_backward(dy, x, params)
        ctx.p16, ctx.grad_dtype, ctx.x32 = p16, grad_dtype, x32
        _, dx, _ = _grid_backward(x32, p16, dy.detach(), None, False, True, False, grad_dtype)
        return dx.to(x.dtype)

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, ddx):
        dy, x, params = ctx.saved_tensors
        v = ddx.detach().float().contiguous()
        gg, dx2, ddy = _grid_backward(ctx.x32, ctx.p16, dy.detach(), v, ctx.needs_input_grad[2],
                                      ctx.needs_input_grad[1], ctx.needs_input_grad[0], ctx.grad_dtype)
        return (ddy.to(dy.dtype) if ddy is not None else None,
                dx2.to(x.dtype) if dx2 is not None else None,
                gg.to(params.dtype) if gg is not None else None, None, None)


class _GridEncode(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, p16, grad_dtype):
        x32 = x.detach().float().contiguous()
        n = x32.shape[0]
        out = torch.empty(n, 32, dtype=torch.float16, device=x.device)
        if n > 0:
            with torch.cuda.device(x.device):
                rc = _lib.lib().gs_grid_encode(_lib.ptr(x32), _lib.ptr(p16), _lib.ptr(out), None, n,
                                               _lib.stream_ptr(x.device))
            _lib.check(rc, "tcnn.Encoding")
        ctx.save_for_backward(x, params)
        ctx.p16, ctx.grad_dtype = p16, grad_dtype
        return out

    @staticmethod
    def backward(ctx, dy):
        x, params = ctx.saved_tensors
        dx = dparams = None
        if ctx.needs_input_grad[0]:
            # differentiable again (create_graph=True): the reference's eikonal / normal terms
            dx = _GridBackwardInput.apply(dy, x, params, ctx.p16, ctx.grad_dtype)
        if ctx.needs_input_grad[1]:
            gg, _, _ = _grid_backward(x.detach().float().contiguous(), ctx.p16, dy.detach(), None, True, False, False,
                                      ctx.grad_dtype)
            dparams = gg.to(params.dtype)
        return dx, dparams, None, None


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
        # table-gradient accumulation: fp32 atomics (default), or torch.float16 = tcnn's own mode (packed fp16
        # atomics under its loss scale of 128)
        self.grad_dtype = torch.float32
        m = _lib.grid_meta() if _have_lib() else None
        total = int(m.total) if m is not None else 6299960
        g = torch.Generator().manual_seed(seed)
        # tcnn initialises grid parameters U(-1e-4, 1e-4)
        self.params = nn.Parameter((torch.rand(total * 2, generator=g) * 2 - 1) * 1e-4)
        self._half = _HalfCache()

    def params_half(self):
        return self._half.get(self.params)

    def forward(self, x, return_dy_dx=False):
        assert x.is_cuda and x.shape[-1] == 3, "tcnn.Encoding runs on the GPU only (no CPU path)"
        if return_dy_dx:        # forward-only helper of the fused pipeline's point queries: analytic d out / d x
            x32 = x.detach().float().contiguous()
            n = x32.shape[0]
            out = torch.empty(n, 32, dtype=torch.float16, device=x.device)
            dy = torch.empty(n, 32, 3, dtype=torch.float32, device=x.device)
            with torch.cuda.device(x.device):
                rc = _lib.lib().gs_grid_encode(_lib.ptr(x32), _lib.ptr(self.params_half()), _lib.ptr(out), _lib.ptr(dy),
                                               n, _lib.stream_ptr(x.device))
            _lib.check(rc, "tcnn.Encoding")
            return out, dy
        return _GridEncode.apply(x, self.params, self.params_half(), self.grad_dtype)


# ---- FullyFusedMLP -------------------------------------------------------------------------------------------------
_FRAG_INDEX = {}


def _mlp_fragment_index(device):
    """Gather indices that turn tcnn's parameter vector (+ one trailing zero) into the 40 MFMA A-fragments of
    gs_mlp_backward ([40,64,8], see include/goslam_neus.h); built once per device."""
    idx = _FRAG_INDEX.get(device)
    if idx is not None:
        return idx
    import numpy as np
    ZERO = 10240
    l = np.arange(64)[:, None]
    e = np.arange(8)[None, :]
    r, kk = (l & 31), 8 * (l >> 5) + e                       # row within the 32-row block, k within the 16-wide step
    frags = []

    def add(n_mt, n_ks, fn):
        for mt in range(n_mt):
            for ks in range(n_ks):
                frags.append(fn(32 * mt + r + 0 * kk, 16 * ks + kk + 0 * r))
    # the contraction index over HIDDEN neurons is enumerated in the order an MFMA accumulator tile holds them (the
    # kernel feeds one GEMM's accumulators to the next as its B fragments): position 16 ks + 8 hf + e <-> neuron
    # 32 (ks >> 1) + 8 (2 (ks & 1) + (e >> 2)) + 4 hf + (e & 3)
    pos = np.arange(64)
    ks_, hf_, e_ = pos >> 4, (pos >> 3) & 1, pos & 7
    hid = 32 * (ks_ >> 1) + 8 * (2 * (ks_ & 1) + (e_ >> 2)) + 4 * hf_ + (e_ & 3)
    add(2, 5, lambda row, k: row * 80 + k)                                     # W1
    add(2, 4, lambda row, k: 5120 + row * 64 + hid[k])                          # W2 (K = H1 neurons, permuted)
    add(2, 1, lambda row, k: 9216 + k * 64 + row)                               # W3^T
    add(2, 4, lambda row, k: 5120 + hid[k] * 64 + row)                          # W2^T (K = dH2 neurons, permuted)
    add(3, 4, lambda row, k: np.where(row < 80, hid[k] * 80 + row, ZERO))       # W1^T (K = dH1 neurons), rows padded to 96
    idx = torch.from_numpy(np.stack(frags).astype(np.int64).reshape(-1)).to(device)
    assert idx.numel() == 40 * 64 * 8
    _FRAG_INDEX[device] = idx
    return idx


_FRAG_INDEX32 = {}


def _mlp_fragment_index32(device):
    """the same indices as int32 (gs_map_step_prep gathers the fragments inside the fused mapper step)"""
    idx = _FRAG_INDEX32.get(device)
    if idx is None:
        idx = _FRAG_INDEX32[device] = _mlp_fragment_index(device).to(torch.int32).contiguous()
    return idx


def _pack_mlp_fragments(W):
    """One gather: [10240] fp16 parameters -> [40,64,8] fp16 fragments."""
    ext = torch.cat([W.reshape(-1), W.new_zeros(1)])
    return ext[_mlp_fragment_index(W.device)].view(40, 64, 8)


class _MLP(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, params, p16, n_in, n_out):
        n = x.shape[0]
        dev = x.device
        # tcnn pads the input to a multiple of 16 with ones; the padded rows are what the backward re-reads
        xp = torch.ones(n, 80, dtype=torch.float16, device=dev)
        xp[:, :n_in] = x.detach()
        out = torch.empty(n, n_out, dtype=torch.float16, device=dev)
        if n > 0:
            with torch.cuda.device(dev):
                rc = _lib.lib().gs_mlp_forward(_lib.ptr(xp), _lib.ptr(p16), _lib.ptr(out), n, 80, n_out, None, 0,
                                               _lib.stream_ptr(dev))
            _lib.check(rc, "tcnn.Network")
        ctx.save_for_backward(x, params)
        ctx.xp, ctx.p16, ctx.n_in, ctx.n_out = xp, p16, n_in, n_out
        return out

    @staticmethod
    @torch.autograd.function.once_differentiable
    def backward(ctx, dout):
        x, params = ctx.saved_tensors
        n, dev = ctx.xp.shape[0], ctx.xp.device
        if n == 0:
            return (torch.zeros_like(x) if ctx.needs_input_grad[0] else None,
                    torch.zeros_like(params) if ctx.needs_input_grad[1] else None, None, None, None)
        L = _lib.lib()
        d3 = torch.zeros(n, 3, dtype=torch.float32, device=dev)
        d3[:, :ctx.n_out] = dout[:, :3] if ctx.n_out >= 3 else dout
        dxp = torch.empty(n, 80, dtype=torch.float16, device=dev)
        nblk = L.gs_mlp_backward_blocks(n)
        partial = torch.empty(nblk, 10240, dtype=torch.float32, device=dev)
        wpack = _pack_mlp_fragments(ctx.p16)
        with torch.cuda.device(dev):
            rc = L.gs_mlp_backward(_lib.ptr(ctx.xp), _lib.ptr(wpack), _lib.ptr(d3), None, LOSS_SCALE, _lib.ptr(dxp),
                                   _lib.ptr(partial), n, _lib.stream_ptr(dev))
        _lib.check(rc, "tcnn.Network backward")
        dx = dparams = None
        if ctx.needs_input_grad[0]:
            dx = (dxp[:, :ctx.n_in].float() / LOSS_SCALE).to(x.dtype)
        if ctx.needs_input_grad[1]:
            dparams = (partial.sum(0) / LOSS_SCALE).to(params.dtype)
        return dx, dparams, None, None, None


class Network(nn.Module):
    """tcnn.Network with otype FullyFusedMLP (InstantNeuS.py:184-192): width 64, 2 hidden layers,
    ReLU, no bias, input padded to a multiple of 16 with ones, output padded to 16."""

    def __init__(self, n_input_dims, n_output_dims, network_config=None, seed=1337):
        super().__init__()
        cfg = dict(network_config or {})
        if cfg.get("n_neurons", 64) != 64 or cfg.get("n_hidden_layers", 2) != 2 or \
                cfg.get("activation", "ReLU") != "ReLU" or cfg.get("output_activation", "none").lower() != "none":
            raise NotImplementedError("only FullyFusedMLP 64x2 ReLU/none (the InstantNeuS colour net) is built")
        if n_input_dims > 80 or n_output_dims > 3:
            raise NotImplementedError("FullyFusedMLP: <= 80 inputs and <= 3 outputs are built")
        self.n_input_dims = n_input_dims
        self.n_output_dims = n_output_dims
        pad_in = 80
        g = torch.Generator().manual_seed(seed)

        def xavier(o, i):
            a = math.sqrt(6.0 / (o + i))
            return ((torch.rand(o, i, generator=g) * 2 - 1) * a).reshape(-1)
        self.params = nn.Parameter(torch.cat([xavier(64, pad_in), xavier(64, 64), xavier(16, 64)]))
        self._half = _HalfCache()

    def params_half(self):
        return self._half.get(self.params)

    def forward(self, x):
        assert x.is_cuda and x.shape[-1] == self.n_input_dims, "tcnn.Network runs on the GPU only (no CPU path)"
        return _MLP.apply(x, self.params, self.params_half(), self.n_input_dims, self.n_output_dims)
