"""The mapper's optimisation step (reference src/mapping.py:60-148, `Mapper.optimize_map`):
render a ray batch, colour / depth / SDF / eikonal losses, backward, clip_grad_norm_(35), AdamW --
with the ray batch optionally sharded over the GPUs of a node (distributed.py)."""
import torch

from .. import _lib
from .distributed import FlatGradReducer, all_reduce_sum_, mapping_loss_sharded, shard_rays


def make_optimizer(model, net_lr=1e-3, grid_lr=1e-2, fused=None):
    """reference src/mapping.py:55-58 (same hyper-parameters).  On the GPU the single-kernel (`fused`) AdamW is
    used: the step is identical, but the 6 parameter tensors are updated by one launch instead of ~20."""
    groups = [{"params": model.get_training_parameters(), "lr": net_lr},
              {"params": model.get_volume_parameters(), "lr": grid_lr}]
    if fused is None:
        fused = all(p.is_cuda for g in groups for p in g["params"])
    kw = dict(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01)
    opt = None
    if fused:
        try:
            opt = torch.optim.AdamW(groups, fused=True, **kw)
        except (RuntimeError, TypeError, ValueError):
            opt = None
    if opt is None:
        opt = torch.optim.AdamW(groups, **kw)
    opt.register_step_post_hook(_bump_versions)
    return opt


def _bump_versions(optimizer, args=None, kwargs=None):
    """torch's single-kernel (`fused=True`) AdamW updates the parameters WITHOUT advancing their version counters
    (measured on torch 2.10 / ROCm: version 0 -> 0, foreach: 0 -> 2).  Everything this package caches per parameter
    version -- the fp16 working copies of the hash table and the colour MLP (tcnn_compat._HalfCache), the host copy of
    the NeuS variance -- would silently keep serving the pre-step values, so optimisers made here advance the
    counters themselves after every step."""
    for g in optimizer.param_groups:
        for p in g["params"]:
            torch.autograd.graph.increment_version(p)


class FlatAdamW:
    """The trained parameters as views of ONE flat fp32 buffer [hash table | colour MLP | sdf_layer.weight | .bias |
    colour _B | variance] with flat AdamW moments and a flat fp16 working copy, stepped by two HIP launches
    (gs_map_grad_sqnorm + gs_map_adamw: global-norm clipping, unscaling of the loss-scaled fp16 table gradient, AdamW
    with the two learning rates of src/mapping.py:55-58, and the fp16 copies the next forward reads).  Same update as
    clip_grad_norm_(35) + torch.optim.AdamW(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01) on the same gradients.
    `sdf_network.encoding._B` is in the reference's parameter list but never receives a gradient, so torch skips it
    (no weight decay either) -- it stays outside the buffer."""

    def __init__(self, model, net_lr=1e-3, grid_lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=35.0):
        net = model.sdf_network
        self.grid_module = net.encoding.encoding
        self.mlp_module = model.color_network.network
        self.grid_p = self.grid_module.params
        self.dense = [("mlp", self.mlp_module.params), ("sdf_w", net.sdf_layer.weight), ("sdf_b", net.sdf_layer.bias),
                      ("cB", model.color_network._B), ("var", model.variance_network.variance)]
        dev = self.grid_p.device
        self.n16 = self.grid_p.numel()
        assert self.n16 % 8 == 0
        sizes = [p.numel() for _, p in self.dense]
        self.n = self.n16 + sum(sizes)
        self.P = torch.empty(self.n, dtype=torch.float32, device=dev)
        off = 0
        self.slices = {}
        with torch.no_grad():
            for name, p in [("grid", self.grid_p)] + self.dense:
                k = p.numel()
                self.P[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = self.P[off:off + k].view(p.shape)        # the module now reads / state_dict()s the flat buffer
                self.slices[name] = (off, off + k)
                off += k
        self.M = torch.zeros_like(self.P)
        self.V = torch.zeros_like(self.P)
        self.P16 = self.P.to(torch.float16)
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.g32 = torch.zeros(self.n - self.n16, dtype=torch.float32, device=dev)
        self.hyper = dict(lr16=grid_lr, lr32=net_lr, b1=betas[0], b2=betas[1], eps=eps, wd=weight_decay, max_norm=max_norm)
        self.steps = 0
        self._publish()

    def _publish(self):
        """hand the fp16 working copies to the modules' caches and invalidate everything keyed on parameter versions"""
        for _, p in [("grid", self.grid_p)] + self.dense:
            torch.autograd.graph.increment_version(p)
        for mod, name in ((self.grid_module, "grid"), (self.mlp_module, "mlp")):
            a, b = self.slices[name]
            p = mod.params
            mod._half._val = self.P16[a:b]
            mod._half._key = (p.data_ptr(), p._version, p.device, p.dtype)

    def dense_grad(self, name):
        a, b = self.slices[name]
        return self.g32[a - self.n16:b - self.n16]

    def step(self, grid_grad16, inv_scale16):
        """grid_grad16: fp16 [n16] = table gradient / inv_scale16; the dense gradients are in self.g32."""
        L = _lib.lib()
        dev = self.P.device
        h = self.hyper
        self.steps += 1
        self.sqnorm.zero_()
        st = _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            _lib.check(L.gs_map_grad_sqnorm(_lib.ptr(grid_grad16), self.n16, inv_scale16, _lib.ptr(self.g32),
                                            self.g32.numel(), _lib.ptr(self.sqnorm), st), "map_grad_sqnorm")
            _lib.check(L.gs_map_adamw(_lib.ptr(self.P), _lib.ptr(self.M), _lib.ptr(self.V), _lib.ptr(self.P16),
                                      _lib.ptr(grid_grad16), self.n16, inv_scale16, _lib.ptr(self.g32), self.n,
                                      h["lr16"], h["lr32"], h["b1"], h["b2"], h["eps"], h["wd"], self.steps,
                                      _lib.ptr(self.sqnorm), h["max_norm"], st), "map_adamw")
        self._publish()


class MapTrainer:
    def __init__(self, model, renderer, net_lr=1e-3, grid_lr=1e-2, w_color=2.0, w_sdf=2.0, w_eikonal=0.1,
                 uncertainty=True, group=None, rank=0, world=1, fused=None):
        """`fused` (default: on for a CUDA model with tiny-cuda-nn's fp16 table gradients): the whole step without an
        autograd graph -- forward, the loss kernel's analytic output gradients, the HIP backward, one flat-buffer
        clip + AdamW -- ~25 launches instead of ~90, no parameter read-back to the host (`step_fused`)."""
        self.model, self.renderer = model, renderer
        self.train_params = model.get_training_parameters() + model.get_volume_parameters()
        self.w = dict(w_color=w_color, w_sdf=w_sdf, w_eikonal=w_eikonal, uncertainty=uncertainty)
        self.group, self.rank, self.world = group, rank, world
        if fused is None:
            fused = (all(p.is_cuda for p in self.train_params) and model.grid_grad_dtype == torch.float16
                     and model.fused_mlp_backward)
        self.fused = bool(fused)
        if self.fused:
            self.flat = FlatAdamW(model, net_lr, grid_lr)
            self.optimizer, self.reducer = None, None
        else:
            self.optimizer = make_optimizer(model, net_lr, grid_lr)
            self.reducer = FlatGradReducer(self.train_params) if world > 1 else None

    def step_fused(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand=None):
        loss, grid16, inv_scale = self.fused_gradients(rays_o, rays_d, rays_color, rays_depth, perturb_rand)
        self.flat.step(grid16, inv_scale)
        return loss

    def fused_gradients(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand=None):
        """forward + loss + HIP backward (+ all-reduce) without an autograd graph.  Returns (global loss, table gradient
        in loss-scaled fp16, its inverse scale); the dense gradients are left in self.flat.g32."""
        from .instant_neus import _neus_backward_raw, _neus_forward_raw
        model, L = self.model, _lib.lib()
        dev = rays_o.device
        f32 = dict(dtype=torch.float32, device=dev)
        if self.world > 1:
            rays_o, rays_d, rays_color, rays_depth = shard_rays([rays_o, rays_d, rays_color, rays_depth],
                                                                self.rank, self.world)
        c = lambda t: t.detach().float().contiguous()
        rays_o, rays_d, rays_color, rays_depth = c(rays_o), c(rays_d), c(rays_color), c(rays_depth).reshape(-1)
        n = rays_o.shape[0]
        z_vals, dists = self.renderer.sample(rays_o, rays_d, model.bound, rays_depth, perturb_rand)
        s = z_vals.shape[1]
        sf = model.variance_network.scale_factor
        var_dev = model.variance_network.variance
        inv_s_dev = torch.exp(var_dev.detach().float() * sf).clamp(1e-6, 1e6).reshape(1)
        color, depth, dvar, normal, wsum, sdf, gerr, zmid, saved = _neus_forward_raw(
            model, rays_o, rays_d, z_vals, dists, 0.0, save=True, inv_s_dev=inv_s_dev)
        # counts over ALL ranks: valid rays (loss normalisation) and rays (eikonal mean)
        counts = torch.stack([(rays_depth > 0).sum().float(), torch.full((), float(n), **f32)])
        if self.world > 1:
            all_reduce_sum_(counts, self.group)
        d_color = torch.empty(n, 3, **f32)
        d_depth = torch.empty(n, 1, **f32)
        d_sdf = torch.empty(n, s, **f32)
        loss_rays = torch.empty(n, **f32)
        w = self.w
        with torch.cuda.device(dev):
            rc = L.gs_mapping_loss(_lib.ptr(color), _lib.ptr(depth), _lib.ptr(dvar), _lib.ptr(sdf), _lib.ptr(zmid),
                                   _lib.ptr(rays_color), _lib.ptr(rays_depth), _lib.ptr(counts[:1]),
                                   float(model.sdf_truncation), float(model.sdf_sparse_factor), float(w["w_color"]),
                                   float(w["w_sdf"]), int(bool(w["uncertainty"])), _lib.ptr(d_color), _lib.ptr(d_depth),
                                   _lib.ptr(d_sdf), _lib.ptr(loss_rays), n, s, _lib.stream_ptr(dev))
        _lib.check(rc, "mapping_loss")
        # eikonal term: w_eik * mean over all points of all ranks -> the same constant for every ray
        d_gerr = (w["w_eikonal"] / (counts[1] * float(s))).expand(n, 1).contiguous()
        g = _neus_backward_raw(model, saved, (rays_o, rays_d, z_vals, dists, sdf, zmid), 0.0, 0.0,
                               d_color, d_depth, None, None, None, d_sdf, d_gerr, inv_s_dev=inv_s_dev, var_dev=var_dev)
        flat = self.flat
        for name in ("mlp", "sdf_w", "sdf_b", "cB", "var"):
            flat.dense_grad(name).copy_(g[name].reshape(-1))
        grid16 = g["grid_acc"]
        if self.world > 1:          # one fp16 collective for the table (25 MB), one small fp32 one for the rest
            all_reduce_sum_(grid16, self.group)
            all_reduce_sum_(flat.g32, self.group)
        loss = loss_rays.sum() + w["w_eikonal"] * gerr.sum() / (counts[1] * float(s))
        if self.world > 1:
            loss = all_reduce_sum_(loss.clone(), self.group)
        return loss, grid16, 1.0 / float(g["grid_scale"])

    def step(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand=None):
        """One joint iteration on the GLOBAL batch (every rank passes the same tensors; each renders
        its contiguous shard).  Returns the global loss as a 0-dim tensor (no host sync in the step)."""
        if self.fused:
            return self.step_fused(rays_o, rays_d, rays_color, rays_depth, perturb_rand)
        if self.world > 1:
            rays_o, rays_d, rays_color, rays_depth = shard_rays([rays_o, rays_d, rays_color, rays_depth],
                                                                self.rank, self.world)
        self.optimizer.zero_grad(set_to_none=False)
        z_vals, dists = self.renderer.sample(rays_o, rays_d, self.model.bound, rays_depth, perturb_rand)
        ret = self.renderer.eval_points(rays_o, rays_d, z_vals, dists, self.model, None)
        loss, loss_value = mapping_loss_sharded(ret, rays_color, rays_depth, self.model.compute_sdf_error,
                                                self.group, **self.w)
        loss.backward()
        if self.reducer is not None:
            self.reducer.reduce(self.group)          # sum of shard gradients == single-GPU gradient
        torch.nn.utils.clip_grad_norm_(self.train_params, max_norm=35.0, foreach=True if rays_o.is_cuda else None)
        self.optimizer.step()
        return loss_value
