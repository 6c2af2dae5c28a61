"""The mapper's optimisation step (reference src/mapping.py:60-148, `Mapper.optimize_map`):
render a ray batch, colour / depth / SDF / eikonal losses, backward, clip_grad_norm_(35), AdamW --
with the ray batch optionally sharded over the GPUs of a node (distributed.py)."""
import torch

from .. import _lib
from .distributed import FlatGradReducer, all_reduce_sum_, mapping_loss_sharded, shard_rays
from .tcnn_compat import _mlp_fragment_index32

MAX_GRAPHS = 8          # captured batch shapes a trainer keeps (the mapper's batches come in a handful of sizes)


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


class HipOptKernels:
    """The optimiser's two HIP launches (csrc/map_opt.hip) behind tensor arguments.  ShardedFlatAdamW takes the object
    as a parameter so that the world-size-2 gloo test can drive the SHARDING logic (slices, padding, collectives) on
    CPU tensors with a torch restatement of the two kernels defined in the test; the product never passes another."""

    def sqnorm(self, out, g16, inv_scale16, g32):
        L = _lib.lib()
        dev = out.device
        n32 = 0 if g32 is None else g32.numel()
        with torch.cuda.device(dev):
            _lib.check(L.gs_map_grad_sqnorm(_lib.ptr(g16), g16.numel(), inv_scale16, _lib.ptr(g32), n32,
                                            _lib.ptr(out), _lib.stream_ptr(dev)), "map_grad_sqnorm")

    def adamw(self, p, m, v, p16, g16, inv_scale16, pd, md, vd, p16d, g32, h, step, step_dev, sqnorm):
        L = _lib.lib()
        dev = p.device
        with torch.cuda.device(dev):
            _lib.check(L.gs_map_adamw_seg(_lib.ptr(p), _lib.ptr(m), _lib.ptr(v), _lib.ptr(p16), _lib.ptr(g16),
                                          g16.numel(), inv_scale16, _lib.ptr(pd), _lib.ptr(md), _lib.ptr(vd),
                                          _lib.ptr(p16d), _lib.ptr(g32), g32.numel(), h["lr16"], h["lr32"], h["b1"],
                                          h["b2"], h["eps"], h["wd"], int(step), _lib.ptr(step_dev), _lib.ptr(sqnorm),
                                          h["max_norm"], _lib.stream_ptr(dev)), "map_adamw")


class FlatAdamW:
    """The trained parameters as views of ONE flat fp32 buffer [hash table (padded) | colour MLP | sdf_layer.weight |
    .bias | colour _B | variance] with flat AdamW moments and a flat fp16 working copy, stepped by two HIP launches
    (gs_map_grad_sqnorm + gs_map_adamw_seg: global-norm clipping, unscaling of the loss-scaled fp16 table gradient,
    AdamW with the two learning rates of src/mapping.py:55-58, and the fp16 copies the next forward reads).  Same update
    as clip_grad_norm_(35) + torch.optim.AdamW(betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01) on the same gradients.
    `sdf_network.encoding._B` is in the reference's parameter list but never receives a gradient, so torch skips it
    (no weight decay either) -- it stays outside the buffer.

    With `world` > 1 the optimiser state of the TABLE is sharded over the ranks (reduce-scatter /
    all-gather over the xGMI mesh instead of all-reduce + a replicated optimiser):

        reduce-scatter(fp16 table gradient)   each rank receives the sum of its 1/G slice        (25.2 MB / G per peer)
        all-reduce(fp32 dense gradients)      46 KB; the global loss rides in a spare slot
        gs_map_grad_sqnorm on the slice       (+ the dense part on rank 0) -> all-reduce of ONE scalar -> global norm
        gs_map_adamw_seg on the slice         fp32 master, m, v of the slice only: the 353 MB pass becomes 353 / G MB;
                                              the 11.5 K dense parameters are stepped identically on every rank
        all-gather(fp16 working copy)         every rank gets the whole updated table for its next forward

    The fp32 master of the slices a rank does not own goes stale between `sync_master()` calls (an all-gather of the
    fp32 slices; call it before state_dict() / checkpoints -- MapTrainer.state_dict does)."""

    def __init__(self, model, net_lr=1e-3, grid_lr=1e-2, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_norm=35.0,
                 rank=0, world=1, group=None, kernels=None, sharded=None):
        net = model.sdf_network
        self.grid_module = net.encoding.encoding
        self.mlp_module = model.color_network.network
        self.grid_p = self.grid_module.params
        self.dense = [("mlp", self.mlp_module.params), ("sdf_w", net.sdf_layer.weight), ("sdf_b", net.sdf_layer.bias),
                      ("cB", model.color_network._B), ("var", model.variance_network.variance)]
        self.rank, self.world, self.group = int(rank), int(world), group
        # the sharded schedule (reduce-scatter -> slice AdamW -> deferred all-gather) is what world > 1 runs; `sharded=True`
        # runs the same schedule in a process group of ONE rank -- how a 1-GPU box takes the RCCL path of this class
        # (tests/test_neus_gpu.py: the collectives become RCCL's own one-rank copies, everything around them is unchanged)
        self.sharded = (self.world > 1) if sharded is None else bool(sharded)
        self.kernels = kernels if kernels is not None else HipOptKernels()
        dev = self.grid_p.device
        self.n16 = self.grid_p.numel()
        assert self.n16 % 8 == 0
        # entries per rank: a multiple of 8 (16-byte aligned fp16 slices); the table is padded to world * slice
        self.slice = -(-self.n16 // (8 * self.world)) * 8
        self.n16p = self.slice * self.world
        self.lo, self.hi = self.rank * self.slice, (self.rank + 1) * self.slice
        self.nd = sum(p.numel() for _, p in self.dense)
        self.n = self.n16 + self.nd                       # trained parameters (without padding)
        self.P = torch.zeros(self.n16p + self.nd, dtype=torch.float32, device=dev)
        off = self.n16p
        self.slices = {"grid": (0, self.n16)}
        with torch.no_grad():
            self.P[:self.n16].copy_(self.grid_p.detach().reshape(-1).float())
            self.grid_p.data = self.P[:self.n16].view(self.grid_p.shape)
            for name, p in self.dense:
                k = p.numel()
                self.P[off:off + k].copy_(p.detach().reshape(-1).float())
                p.data = self.P[off:off + k].view(p.shape)      # the module now reads / state_dict()s the flat buffer
                self.slices[name] = (off, off + k)
                off += k
        # moments: the own table slice + the dense parameters
        self.M = torch.zeros(self.slice + self.nd, dtype=torch.float32, device=dev)
        self.V = torch.zeros_like(self.M)
        self.P16 = self.P.to(torch.float16)
        self.G16 = torch.zeros(self.n16p, dtype=torch.float16, device=dev)        # the backward's table-gradient buffer
        self.g16s = torch.empty(self.slice, dtype=torch.float16, device=dev) if self.sharded else None
        self.sqnorm = torch.zeros(1, dtype=torch.float32, device=dev)
        self.g32 = torch.zeros(self.nd + 2, dtype=torch.float32, device=dev)      # dense gradients | global loss | spare
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.hyper = dict(lr16=grid_lr, lr32=net_lr, b1=betas[0], b2=betas[1], eps=eps, wd=weight_decay, max_norm=max_norm)
        self.steps = 0
        self.overlap_gather = True          # world > 1: the all-gather of the updated table is waited for by its next reader
        self._gather_wait = None
        self._rs_wait = None                # world > 1: the reduce-scatter started under the backward's last kernels
        self._exchange_timer = None         # bench.py: ExchangeTimer (events around the collectives on the step's stream)
        self._publish()

    # the reference's mapper scales its learning rates per iteration (src/mapping.py: lr_factor) through param_groups
    @property
    def param_groups(self):
        return [{"name": "network", "lr": self.hyper["lr32"]}, {"name": "volume", "lr": self.hyper["lr16"]}]

    def set_lr(self, net_lr=None, grid_lr=None):
        if net_lr is not None:
            self.hyper["lr32"] = float(net_lr)
        if grid_lr is not None:
            self.hyper["lr16"] = float(grid_lr)

    def _publish(self):
        """hand the fp16 working copies to the modules' caches and invalidate everything keyed on parameter versions"""
        for _, p in [("grid", self.grid_p)] + self.dense:
            torch.autograd.graph.increment_version(p)
        for mod, name in ((self.grid_module, "grid"), (self.mlp_module, "mlp")):
            a, b = self.slices[name]
            p = mod.params
            mod._half._val = self.P16[a:b]
            mod._half._key = (p.data_ptr(), p._version, p.device, p.dtype)

    def check_bindings(self):
        """the modules' parameters must still be views of the flat buffer (a later `.half()`, `.to(device)`,
        `share_memory()` or deepcopy rebinds `.data`: the step would then update memory nobody reads)"""
        base = self.P.data_ptr()
        for name, p in [("grid", self.grid_p)] + self.dense:
            a, _ = self.slices[name]
            if p.data_ptr() != base + 4 * a:
                raise RuntimeError(f"FlatAdamW: parameter '{name}' no longer lives in the flat buffer (the model was "
                                   "moved / converted / copied after the trainer was built); build a new MapTrainer")

    def dense_grad(self, name):
        a, b = self.slices[name]
        return self.g32[a - self.n16p:b - self.n16p]

    def grad_table(self):
        """the (zeroed) loss-scaled fp16 table-gradient buffer the backward accumulates into"""
        self.G16.zero_()
        return self.G16[:self.n16]

    def step(self, inv_scale16, prepped=False, captured=False):
        """table gradient / inv_scale16 is in self.G16 (fp16), the dense gradients (+ the loss in slot nd) in self.g32.
        `prepped`: gs_map_step_prep already advanced the device-side step count and zeroed the norm accumulator.
        `captured`: the call is being captured into a hipGraph together with its collectives (MapTrainer's one-graph
        sharded step): every collective is stream-ordered (no deferred all-gather, no timer events) and the host-side
        bookkeeping (`_publish`) is the caller's, once per replay."""
        from .distributed import all_gather_into_, all_reduce_sum_, reduce_scatter_sum_
        K, h, nd = self.kernels, self.hyper, self.nd
        self.steps += 1
        if not prepped:
            self.step_dev.add_(1)
            self.sqnorm.zero_()
        dense = (self.P[self.n16p:], self.M[self.slice:], self.V[self.slice:], self.P16[self.n16p:], self.g32[:nd])
        if self.sharded:
            if not captured:
                self.wait_gather()
            t = None if captured else self._exchange_timer
            if t is not None:
                t.mark("rs0")
            early, self._rs_wait = self._rs_wait, None
            if early is None:
                reduce_scatter_sum_(self.g16s, self.G16, self.group)
            all_reduce_sum_(self.g32, self.group)
            if early is not None:       # started under the Gram / post kernels (begin_reduce_scatter): enqueued before the
                early()                 # dense all-reduce on the one communicator, so it is complete by now
            if t is not None:
                t.mark("rs1")
            K.sqnorm(self.sqnorm, self.g16s, inv_scale16, self.g32[:nd] if self.rank == 0 else None)
            if t is not None:
                t.mark("n0")
            all_reduce_sum_(self.sqnorm, self.group)
            if t is not None:
                t.mark("n1")
            K.adamw(self.P[self.lo:self.hi], self.M[:self.slice], self.V[:self.slice], self.P16[self.lo:self.hi],
                    self.g16s, inv_scale16, *dense, h, self.steps, self.step_dev, self.sqnorm)
            # the updated fp16 table travels while the host prepares the next step (input copies, the batch's counts):
            # the all-gather is only enqueued here; whoever reads the table next -- the next step (wait_gather), a render
            # between steps (the cache's `_pending` hook), sync_master -- waits for it first.  No other collective is
            # issued in between, so every rank enqueues the same sequence on the one communicator.
            if t is not None:
                t.mark("agq")
            self._gather_wait = all_gather_into_(self.P16[:self.n16p], self.P16[self.lo:self.hi], self.group,
                                                 async_op=self.overlap_gather and not captured)
            if captured or not self.overlap_gather:
                self._gather_wait = None
                if t is not None:
                    t.mark("agw0", at="agq")
                    t.mark("agw1")
        else:
            K.sqnorm(self.sqnorm, self.G16, inv_scale16, self.g32[:nd])
            K.adamw(self.P[:self.n16p], self.M[:self.slice], self.V[:self.slice], self.P16[:self.n16p], self.G16,
                    inv_scale16, *dense, h, self.steps, self.step_dev, self.sqnorm)
        if captured:
            return
        self._publish()
        if self._gather_wait is not None:
            self.grid_module._half._pending = self.wait_gather

    def begin_reduce_scatter(self):
        """start the reduce-scatter of the fp16 table gradient NOW (it is complete: the bin reduce has run) instead of at
        the head of step(): the Gram and post kernels of the backward then run beside it.  Every rank calls this at the same
        point of its step, so the collectives are enqueued in the same order everywhere."""
        if self.sharded and self._rs_wait is None:
            from .distributed import reduce_scatter_sum_
            self._rs_wait = reduce_scatter_sum_(self.g16s, self.G16, self.group, async_op=True)

    def wait_gather(self):
        """make the current stream wait for the deferred all-gather of the fp16 table (no-op when none is in flight)"""
        w, self._gather_wait = self._gather_wait, None
        self.grid_module._half._pending = None
        if w is not None:
            t = self._exchange_timer
            if t is not None:
                t.mark("agw0")
            w()
            if t is not None:
                t.mark("agw1")

    def sync_master(self):
        """all-gather the fp32 master slices so that every rank's `P` (= the modules' parameters, state_dict()) is whole"""
        if self.sharded:
            from .distributed import all_gather_into_
            self.wait_gather()
            all_gather_into_(self.P[:self.n16p], self.P[self.lo:self.hi], self.group)

    def collective_bytes(self):
        """bytes a rank sends per step: reduce-scatter + all-gather of the fp16 table, the dense all-reduce, one scalar"""
        if self.world == 1:
            return 0
        f = (self.world - 1) / self.world
        return int(2 * f * 2 * self.n16p + 2 * f * 4 * self.g32.numel() + 8)


class MapTrainer:
    def __init__(self, model, renderer, net_lr=1e-3, grid_lr=1e-2, w_color=2.0, w_sdf=2.0, w_eikonal=0.1,
                 uncertainty=True, group=None, rank=0, world=1, fused=None, graph=None, sharded=None):
        """`fused` (default: on for a CUDA model with tiny-cuda-nn's fp16 table gradients): the whole step without an
        autograd graph -- forward, the loss kernel's analytic output gradients, the HIP backward, one flat-buffer
        clip + AdamW -- no parameter read-back to the host (`step_fused`).  `graph` (default: on with `fused`): the
        step's launch sequence is captured once per batch size in a hipGraph and replayed (one graph launch instead of
        ~65 kernel launches; with world > 1 the collectives stay outside: [sample + forward + loss + backward] is one
        graph, the optimiser's two kernels sit between the collectives).  `sharded=True` runs the world > 1 schedule
        (global counts outside the graph, two graphs around the early reduce-scatter, sharded optimiser, deferred
        all-gather) in a process group of one rank: the RCCL path of the step on a 1-GPU box."""
        self.model, self.renderer = model, renderer
        self.train_params = model.get_training_parameters() + model.get_volume_parameters()
        self.w = dict(w_color=w_color, w_sdf=w_sdf, w_eikonal=w_eikonal, uncertainty=uncertainty)
        self.group, self.rank, self.world = group, rank, world
        self.sharded = (world > 1) if sharded is None else bool(sharded)
        if fused is None:
            fused = all(p.is_cuda for p in self.train_params) and model.grid_grad_dtype == torch.float16
        self.fused = bool(fused)
        self.graph = bool(self.fused if graph is None else (graph and self.fused))
        # sharded step as ONE hipGraph with its collectives captured (RCCL enqueues are graph nodes: no host-side launch
        # between the two halves of the step, no optimiser launches outside a graph): tried on the first capture when the
        # group's backend is RCCL, given up for good (two graphs around eager collectives, as rounds 4-5) if the capture is
        # refused.  GOSLAM_CAPTURE_COLLECTIVES=0 turns the attempt off.
        import os
        self.capture_collectives = os.environ.get("GOSLAM_CAPTURE_COLLECTIVES", "1") == "1"
        self.capture_collectives_error = None
        self._graphs, self._bufs = {}, {}
        if self.fused:
            self.flat = FlatAdamW(model, net_lr, grid_lr, rank=rank, world=world, group=group, sharded=self.sharded)
            self.optimizer, self.reducer = self.flat, None        # `.param_groups` / `.set_lr` facade
        else:
            self.optimizer = make_optimizer(model, net_lr, grid_lr)
            self.reducer = FlatGradReducer(self.train_params) if world > 1 else None

    def state_dict(self):
        """the model's state dict with whole fp32 masters on every rank (sharded optimiser: gathers the slices first)"""
        if self.fused:
            self.flat.sync_master()
        return self.model.state_dict()

    # ---- the fused step ------------------------------------------------------------------------------------------
    def _step_buffers(self, n, dev):
        """persistent per-batch-size scratch of the step.  A captured graph keeps raw pointers into these tensors, so
        every graph entry holds a reference to the dict it was captured with (`ent["bufs"]`): evicting a size from this
        cache can then never free memory a replay still reads or writes."""
        b = self._bufs.get(n)
        if b is None:
            f32 = dict(dtype=torch.float32, device=dev)
            b = dict(counts=torch.zeros(3, **f32), inv_s=torch.zeros(1, **f32), d_gerr=torch.zeros(max(n, 1), 1, **f32),
                     d_invs=torch.zeros(1, **f32), zeros_n1=torch.zeros(n, 1, **f32), zeros_n3=torch.zeros(n, 3, **f32),
                     sdf_wt=torch.zeros(16 * 2 * 32, **f32),
                     mlp_wpack=torch.zeros(40, 64, 8, dtype=torch.float16, device=dev))
            if len(self._bufs) >= 2 * MAX_GRAPHS:
                self._bufs.pop(next(iter(self._bufs)))
            self._bufs[n] = b
        return b

    def _rt_bound_dev(self, dev):
        """the model's `realtime_bound` buffer as the kernels' device-side bound: `InstantNeuS.update_bound` rewrites
        it in place (src/InstantNeuS.py:255-257), so eager steps and replays of a captured graph both mask with the
        CURRENT bound (src/InstantNeuS.py:310) -- a host copy passed by value would be frozen at capture time."""
        rb = self.model.realtime_bound
        if rb.device != dev or rb.dtype != torch.float32 or not rb.is_contiguous():
            raise RuntimeError("MapTrainer: model.realtime_bound must be a contiguous fp32 buffer on the rays' device")
        return rb

    def _local_gradients(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand, counts, bufs=None, after_table=None):
        """THIS RANK's rays: sample + forward + loss kernel + HIP backward, no autograd graph, no collective, no host
        sync.  `counts` = [valid rays, rays, max depth] over ALL ranks (device fp32[3]) or None (single rank: computed
        here).  Leaves the loss-scaled fp16 table gradient in flat.G16, the dense gradients in flat.g32[:nd] and this
        rank's share of the loss in flat.g32[nd]; also zeroes the optimiser's norm accumulator and advances its
        device-side step count (gs_map_step_prep), so flat.step must be called with prepped=True."""
        from .instant_neus import _neus_backward_raw, _neus_forward_raw
        model, L, flat = self.model, _lib.lib(), self.flat
        dev = rays_o.device
        f32 = dict(dtype=torch.float32, device=dev)
        n = rays_o.shape[0]
        w = self.w
        s = self.renderer.N_samples + self.renderer.N_surface
        B = bufs if bufs is not None else self._step_buffers(n, dev)
        sf = float(model.variance_network.scale_factor)
        var_dev = model.variance_network.variance
        st = _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            _lib.check(L.gs_map_step_prep(_lib.ptr(rays_depth), n, _lib.ptr(var_dev), sf, float(w["w_eikonal"]), s,
                                          _lib.ptr(counts), _lib.ptr(B["counts"]), _lib.ptr(B["inv_s"]),
                                          _lib.ptr(B["d_gerr"]), _lib.ptr(B["d_invs"]), _lib.ptr(flat.sqnorm),
                                          _lib.ptr(flat.step_dev), _lib.ptr(model.sdf_network.sdf_layer.weight),
                                          _lib.ptr(B["sdf_wt"]), _lib.ptr(model.color_network.network.params_half()),
                                          _lib.ptr(_mlp_fragment_index32(dev)), _lib.ptr(B["mlp_wpack"]), st),
                       "map_step_prep")
        counts, inv_s_dev = B["counts"], B["inv_s"]
        z_vals, dists = self.renderer.sample(rays_o, rays_d, model.bound, rays_depth, perturb_rand,
                                             gt_max_dev=counts[2:3])
        assert z_vals.shape[1] == s
        color, depth, dvar, normal, wsum, sdf, gerr, zmid, saved = _neus_forward_raw(
            model, rays_o, rays_d, z_vals, dists, 0.0, save=True, inv_s_dev=inv_s_dev,
            rt_bound_dev=self._rt_bound_dev(dev))
        d_color = torch.empty(n, 3, **f32)
        d_depth = torch.empty(n, 1, **f32)
        d_sdf = torch.empty(n, s, **f32)
        loss_rays = torch.empty(n, **f32)
        with torch.cuda.device(dev):
            rc = L.gs_mapping_loss(_lib.ptr(color), _lib.ptr(depth), _lib.ptr(dvar), _lib.ptr(sdf), _lib.ptr(zmid),
                                   _lib.ptr(rays_color), _lib.ptr(rays_depth), _lib.ptr(counts[:1]),
                                   float(model.sdf_truncation), float(model.sdf_sparse_factor), float(w["w_color"]),
                                   float(w["w_sdf"]), int(bool(w["uncertainty"])), _lib.ptr(d_color), _lib.ptr(d_depth),
                                   _lib.ptr(d_sdf), _lib.ptr(loss_rays), n, s, st)
        _lib.check(rc, "mapping_loss")
        g = _neus_backward_raw(model, saved, (rays_o, rays_d, z_vals, dists, sdf, zmid), 0.0, 0.0,
                               d_color, d_depth, None, None, None, d_sdf, B["d_gerr"][:n], inv_s_dev=inv_s_dev,
                               var_dev=var_dev, grid_acc_out=flat.grad_table(), raw_dense=B, after_table=after_table)
        gram, part = g["gram"], g["mlp_partial"]
        with torch.cuda.device(dev):
            _lib.check(L.gs_map_step_post(_lib.ptr(gram), gram.shape[0], 1.0 / float(g["loss_scale"]), _lib.ptr(part),
                                          part.shape[0], _lib.ptr(B["d_invs"]), _lib.ptr(var_dev), _lib.ptr(inv_s_dev), sf,
                                          _lib.ptr(loss_rays), _lib.ptr(gerr), n, float(w["w_eikonal"]), s,
                                          _lib.ptr(counts), _lib.ptr(flat.g32), st), "map_step_post")
        return 1.0 / float(g["grid_scale"])

    def _prepare(self, rays_o, rays_d, rays_color, rays_depth):
        if self.world > 1:
            rays_o, rays_d, rays_color, rays_depth = shard_rays([rays_o, rays_d, rays_color, rays_depth],
                                                                self.rank, self.world)
        c = lambda t: t.detach().float().contiguous()
        return c(rays_o), c(rays_d), c(rays_color), c(rays_depth).reshape(-1)

    def _counts(self, rays_depth):
        """None on a single rank (gs_map_step_prep computes them inside the step); otherwise [valid rays, rays, max
        depth] over the WHOLE batch: the loss normalisers (means over VALID rays, src/mapping.py:96-121) and the
        batch-wide depth maximum the sampler clamps with (src/render.py:121,140).  `rays_depth` is the GLOBAL batch's
        depth column -- every rank is handed the same batch and renders its shard of it -- so the three numbers are
        local reductions: no collective (and no collective latency) stands at the head of a step."""
        if not self.sharded:
            return None
        rd = rays_depth.detach().float().reshape(-1)
        dev = rd.device
        n = rd.shape[0]
        mx = rd.max() if n else torch.zeros((), dtype=torch.float32, device=dev)
        return torch.stack([(rd > 0).sum().float(), torch.full((), float(n), dtype=torch.float32, device=dev), mx.float()])

    def fused_gradients(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand=None):
        """forward + loss + HIP backward without an autograd graph and WITHOUT the optimiser's collectives.  Returns
        (this rank's loss share, the rank-local table gradient in loss-scaled fp16, its inverse scale); the dense
        gradients are left in self.flat.g32.  (Tests / tools; `step_fused` is the production entry.)"""
        counts = self._counts(rays_depth)
        rays_o, rays_d, rays_color, rays_depth = self._prepare(rays_o, rays_d, rays_color, rays_depth)
        self.flat.wait_gather()
        inv_scale = self._local_gradients(rays_o, rays_d, rays_color, rays_depth, perturb_rand, counts)
        self.flat.step_dev.sub_(1)          # (no optimiser step follows: undo gs_map_step_prep's step count)
        return self.flat.g32[self.flat.nd].clone(), self.flat.G16[:self.flat.n16], inv_scale

    def _graph_for(self, args, counts, perturb_rand):
        """hipGraph of the step's local part for this batch shape: static input buffers + the captured launch sequence"""
        h = self.flat.hyper
        key = (tuple(args[0].shape), perturb_rand is None, self.world, h["lr16"], h["lr32"],     # (host scalars are baked in)
               self._one_graph())
        ent = self._graphs.get(key)
        if ent is not None:
            self._graphs[key] = self._graphs.pop(key)       # most recently used last
            return ent
        if len(self._graphs) >= MAX_GRAPHS:                 # evict the least recently used graph (dict order = use order)
            old = self._graphs.pop(next(iter(self._graphs)))
            old["graph"] = old["tail"] = None
        dev = args[0].device
        static = [torch.empty_like(a) for a in args]
        s_counts = torch.zeros(3, dtype=torch.float32, device=dev) if counts is None else torch.empty_like(counts)
        s_pr = None if perturb_rand is None else torch.empty_like(perturb_rand.detach().float().contiguous())
        ent = dict(static=static, counts=s_counts, pr=s_pr, graph=None, tail=None, inv_scale=None, warm=0,
                   bufs=self._step_buffers(args[0].shape[0], dev), rt_bound=self.model.realtime_bound)
        self._graphs[key] = ent
        return ent

    def _one_graph(self):
        """the sharded step's collectives are captured with it (RCCL process group, no exchange timer attached)"""
        if not (self.sharded and self.capture_collectives and self.flat._exchange_timer is None):
            return False
        import torch.distributed as dist
        try:
            return dist.is_available() and dist.is_initialized() and dist.get_backend(self.group) == "nccl"
        except (RuntimeError, ValueError):
            return False

    def step_fused(self, rays_o, rays_d, rays_color, rays_depth, perturb_rand=None):
        flat = self.flat
        flat.check_bindings()
        counts = self._counts(rays_depth)
        args = self._prepare(rays_o, rays_d, rays_color, rays_depth)
        early = flat.begin_reduce_scatter if self.sharded else None       # (the reduce-scatter starts under Gram + post)
        if not self.graph or args[0].shape[0] == 0:
            flat.wait_gather()
            inv_scale = self._local_gradients(*args, perturb_rand, counts, after_table=early)
            flat.step(inv_scale, prepped=True)
            return self._global_loss()
        if perturb_rand is None and self.renderer.perturb > 0:      # drawn OUTSIDE the graph: a replay must see new values
            perturb_rand = self.renderer._perturb_row(self.renderer.N_samples, args[0].device)
        ent = self._graph_for(args, counts, perturb_rand)
        if ent["rt_bound"].data_ptr() != self.model.realtime_bound.data_ptr():
            raise RuntimeError("MapTrainer: model.realtime_bound was re-allocated after a step was captured "
                               "(model moved / copied); build a new MapTrainer")
        dsts, srcs = list(ent["static"]), list(args)
        if counts is not None:
            dsts.append(ent["counts"])
            srcs.append(counts)
        if ent["pr"] is not None:
            dsts.append(ent["pr"])
            srcs.append(perturb_rand.detach().float().contiguous())
        torch._foreach_copy_(dsts, srcs)    # ONE launch for the 4-6 input tensors (was a 4.4 us copy kernel each)
        flat.wait_gather()                  # (world > 1) the previous step's table all-gather ran beside everything above
        whole = not self.sharded            # single GPU: the optimiser's two launches are part of the graph
        one = self._one_graph()             # sharded, collectives captured: also ONE graph

        def body(after_table=None, captured=False):
            inv = self._local_gradients(*ent["static"], ent["pr"], None if whole else ent["counts"], bufs=ent["bufs"],
                                        after_table=after_table)
            if whole or captured:
                flat.step(inv, prepped=True, captured=captured)
            return inv
        if ent["graph"] is None:
            if ent["warm"] < 2:             # eager first (workspaces, fp16 caches, lazy library state), then capture
                ent["warm"] += 1
                inv = body(early)
                if not whole:
                    flat.step(inv, prepped=True)
                return self._global_loss()
            steps_before = flat.steps
            graph = torch.cuda.CUDAGraph()
            # thread-local capture: with world > 1 RCCL's watchdog thread polls events while this thread captures, which a
            # global-mode capture would treat as an illegal call
            if whole:
                with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                    ent["inv_scale"] = body()
            elif one:
                try:
                    with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                        ent["inv_scale"] = body(early, captured=True)
                    ent["one"] = True
                except Exception as exc:        # noqa: BLE001 -- RCCL / torch refused: eager collectives from now on
                    self.capture_collectives, self.capture_collectives_error = False, repr(exc)[:300]
                    flat._rs_wait = None
                    flat.steps = steps_before
                    torch.cuda.synchronize(ent["static"][0].device)
                    self._graphs.pop(next(k for k, v in self._graphs.items() if v is ent))
                    inv = self._local_gradients(*args, perturb_rand, counts, after_table=early)
                    flat.step(inv, prepped=True)
                    return self._global_loss()
            else:
                # world > 1: TWO graphs sharing one memory pool, cut where the table gradient is complete (after the bin
                # reduce): [sample ... backward pass 1 + bin reduce] | [Gram + post].  The reduce-scatter is enqueued
                # between their replays and runs beside the second.
                tail = torch.cuda.CUDAGraph()
                dev = ent["static"][0].device
                torch.cuda.synchronize(dev)
                cap = torch.cuda.Stream(device=dev)
                cap.wait_stream(torch.cuda.current_stream(dev))
                with torch.cuda.stream(cap):
                    graph.capture_begin(capture_error_mode="thread_local")

                    def cut():
                        graph.capture_end()
                        tail.capture_begin(pool=graph.pool(), capture_error_mode="thread_local")
                    ent["inv_scale"] = body(cut)
                    tail.capture_end()
                torch.cuda.current_stream(dev).wait_stream(cap)
                ent["tail"] = tail
            ent["graph"] = graph
            flat.steps = steps_before       # capture ran nothing: the replay below is the step
        ent["graph"].replay()
        if whole or ent.get("one"):
            flat.steps += 1                 # (the device-side count was advanced inside the graph)
            flat._publish()
        else:
            flat.begin_reduce_scatter()
            ent["tail"].replay()
            flat.step(ent["inv_scale"], prepped=True)
        return self._global_loss()

    def _global_loss(self):
        """the step's loss over all ranks as a 0-dim tensor (with world > 1 it was summed by the dense all-reduce)"""
        return self.flat.g32[self.flat.nd].clone()

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
