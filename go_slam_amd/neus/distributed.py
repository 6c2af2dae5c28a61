"""Data-parallel mapping across the GPUs of one node: the mapper's ray batch is
sharded over ranks, every rank holds a full replica of the 12.6 M-entry grid and the MLPs, and one
exchange step per iteration sums the gradients over RCCL / xGMI.  The production (fused) trainer shards the OPTIMISER
too: reduce-scatter of the fp16 table gradient, AdamW on the owned 1/G slice, all-gather of the fp16 working copy
(neus/mapper.py: FlatAdamW); the all-reduce of one flat fp32 buffer described below serves the autograd path.

The reference has no distributed code at all (its only multi-device knob is mapping.device); DDP
cannot be used because the loss needs `autograd.grad` (InstantNeuS.py:139), so the exchange is an
explicit all-reduce of ONE flat fp32 buffer (50.4 MB: a single large collective suits xGMI's
per-link-bound rings better than per-parameter calls).  For single-GPU parity the loss terms,
which are means over *valid* rays, are normalised with GLOBAL counts, so the SUM of the rank
gradients is exactly the gradient of the single-GPU loss.
"""
import torch
import torch.distributed as dist


def shard_bounds(n, rank, world):
    """Contiguous, near-even split of n rays: [lo, hi) for `rank`."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_rays(tensors, rank, world):
    lo, hi = shard_bounds(tensors[0].shape[0], rank, world)
    return [t[lo:hi] for t in tensors]


def all_reduce_sum_(t, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def _gloo_on_device(t, group):
    """functional smoke runs put several ranks on ONE GPU, where RCCL refuses to start; gloo then carries the
    collectives, and its reduce-scatter / all-gather take host tensors only"""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def reduce_scatter_sum_(out, inp, group=None, async_op=False):
    """out[i] = sum over ranks of inp[rank * len(out) + i] -- RCCL reduce-scatter over xGMI: every rank ends up with the
    summed 1/G slice it owns (the sharded optimiser's input), moving (G-1)/G of the buffer once instead of twice.
    `async_op`: only enqueued (on RCCL's stream, behind what the current stream has issued so far); the returned callable
    makes the current stream wait for it."""
    if _gloo_on_device(inp, group):
        h = torch.empty(out.shape, dtype=out.dtype)
        work = dist.reduce_scatter_tensor(h, inp.cpu(), op=dist.ReduceOp.SUM, group=group, async_op=async_op)
        if async_op:
            def finish():
                work.wait()
                out.copy_(h)
            return finish
        out.copy_(h)
        return out
    work = dist.reduce_scatter_tensor(out, inp, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work.wait if async_op else out


def all_gather_into_(full, part, group=None, async_op=False):
    """full = concatenation over ranks of `part` (in place when `part` is this rank's slice of `full`).  `async_op`: the
    collective is only ENQUEUED (on RCCL's own stream, behind what the current stream has issued so far); the returned
    callable makes the current stream wait for it -- kernels launched in between run beside the transfer."""
    if _gloo_on_device(full, group):
        h = torch.empty(full.shape, dtype=full.dtype)
        work = dist.all_gather_into_tensor(h, part.cpu(), group=group, async_op=async_op)
        if async_op:
            def finish():
                work.wait()
                full.copy_(h)
            return finish
        full.copy_(h)
        return full
    work = dist.all_gather_into_tensor(full, part, group=group, async_op=async_op)
    return work.wait if async_op else full


class ExchangeTimer:
    """Events on the step's stream around the sharded optimiser's collectives (FlatAdamW.step marks them when a timer is
    attached): the time the compute stream spends BLOCKED on an exchange -- reduce-scatter + dense all-reduce (rs0..rs1),
    the clip norm's scalar all-reduce (n0..n1) and the wait for the table all-gather (agw0..agw1; with the deferred
    all-gather that wait sits at the head of the NEXT step, behind the host-side preparation it overlaps with).  What a
    multi-GPU bench line reports next to the step time so that a scaling run diagnoses itself."""

    # (rs0 is marked where the stream starts to WAIT for the reduce-scatter: with the early start -- under the Gram / post
    # kernels -- that is the dense all-reduce behind them, not the point where the collective was enqueued)
    PAIRS = (("rs0", "rs1", "reduce_scatter_and_dense_allreduce"), ("n0", "n1", "clip_norm_allreduce"),
             ("agw0", "agw1", "all_gather_wait"))

    def __init__(self, device):
        self.device = device
        self.marks = []

    def mark(self, name, at=None):
        if at is not None:                          # alias of an event already recorded
            ev = next(e for n, e in reversed(self.marks) if n == at)
        else:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream(self.device))
        self.marks.append((name, ev))

    def reset(self):
        self.marks = []

    def exposed_ms(self):
        """{phase: total ms the stream was blocked in it} over everything marked since reset()"""
        torch.cuda.synchronize(self.device)
        out = {k: 0.0 for _, _, k in self.PAIRS}
        for a, b, key in self.PAIRS:
            start = None
            for n, e in self.marks:
                if n == a:
                    start = e
                elif n == b and start is not None:
                    out[key] += start.elapsed_time(e)
                    start = None
        out["total"] = sum(out.values())
        return out


class _MapLossFn(torch.autograd.Function):
    """Colour + depth + SDF terms of the mapper's loss and their gradients in one HIP launch (gs_mapping_loss)."""

    @staticmethod
    def forward(ctx, color, depth, dvar, sdf, z_vals, rays_color, rays_depth, counts, trunc, sparse, w_color, w_sdf,
                uncertainty):
        from .. import _lib
        n, s = sdf.shape
        f = lambda t: t.detach().float().contiguous()
        color, depth, dvar, sdf, z_vals = f(color), f(depth), f(dvar), f(sdf), f(z_vals)
        rays_color, rays_depth, counts = f(rays_color), f(rays_depth), f(counts)
        d_color = torch.empty_like(color)
        d_depth = torch.empty(n, 1, dtype=torch.float32, device=sdf.device)
        d_sdf = torch.empty_like(sdf)
        loss_rays = torch.empty(n, dtype=torch.float32, device=sdf.device)
        with torch.cuda.device(sdf.device):
            rc = _lib.lib().gs_mapping_loss(_lib.ptr(color), _lib.ptr(depth), _lib.ptr(dvar), _lib.ptr(sdf),
                                            _lib.ptr(z_vals), _lib.ptr(rays_color), _lib.ptr(rays_depth),
                                            _lib.ptr(counts), float(trunc), float(sparse), float(w_color), float(w_sdf),
                                            int(bool(uncertainty)), _lib.ptr(d_color), _lib.ptr(d_depth),
                                            _lib.ptr(d_sdf), _lib.ptr(loss_rays), n, s, _lib.stream_ptr(sdf.device))
        _lib.check(rc, "mapping_loss")
        ctx.save_for_backward(d_color, d_depth, d_sdf)
        return loss_rays.sum()

    @staticmethod
    def backward(ctx, g):
        d_color, d_depth, d_sdf = ctx.saved_tensors
        return (d_color * g, d_depth * g, None, d_sdf * g) + (None,) * 9


def mapping_loss_sharded(ret, rays_color, rays_depth, compute_sdf_error, group=None, w_color=2.0, w_sdf=2.0,
                         w_eikonal=0.1, uncertainty=True, fused=True):
    """Mapper.optimize_map's loss (reference src/mapping.py:96-132) on this rank's ray shard,
    normalised by global counts.  `compute_sdf_error(sdf, z_vals, gt_depth)` is the model's
    (InstantNeuS.py:372-400).  Returns (local_loss, global_loss as a 0-dim tensor): local_loss.backward()
    followed by an all-reduce(SUM) of the gradients reproduces the single-GPU gradient."""
    # Sync-free formulation: the reference gathers the valid rays with boolean indexing (`x[mask]`: a nonzero
    # + host sync per use, ~25 of them with their backward index_puts); every term is a masked sum divided by
    # a count, so the same numbers come from multiplying by the 0/1 mask -- no gather, no host round trip.
    # At 4096 rays per GPU (8-way sharding of a 32768-ray batch) the gathers were a quarter of the step.
    rd = rays_depth.reshape(-1, 1)
    vmf = (rd > 0).to(rd.dtype)                                    # [n,1] 1 for rays with a depth measurement
    n_local = torch.stack([vmf.sum().double(), torch.tensor(float(rd.shape[0]), dtype=torch.float64, device=rd.device)])
    n_glob = all_reduce_sum_(n_local.clone(), group)
    nv_l, nr_l = n_local[0].to(rd.dtype), n_local[1].to(rd.dtype)
    nv_g, nr_g = n_glob[0].to(rd.dtype), n_glob[1].to(rd.dtype)
    dv = ret["depth_variance"]
    model = getattr(compute_sdf_error, "__self__", None)
    if rd.is_cuda and fused and model is not None and ret["sdf"].shape[1] <= 128:
        # one launch for the colour / depth / SDF terms and their gradients; only the eikonal mean stays in torch
        total = _MapLossFn.apply(ret["color"], ret["depth"], dv, ret["sdf"], ret["z_vals"], rays_color, rd,
                                 n_glob[:1].float(), model.sdf_truncation, model.sdf_sparse_factor, w_color, w_sdf,
                                 uncertainty)
        total = total + w_eikonal * ret["gradient_error"].mean() * (nr_l / nr_g)
        glob = all_reduce_sum_(total.detach().clone().double(), group)
        return total, glob
    uw = 1.0 / torch.sqrt(dv.detach() + 1e-10) if uncertainty else torch.ones_like(dv)
    total = (torch.abs(ret["color"] - rays_color) * vmf).sum() / (3.0 * nv_g) * w_color
    total = total + (torch.abs(ret["depth"] - rd) * uw * vmf).sum() / nv_g
    e, f = compute_sdf_error(ret["sdf"], ret["z_vals"], rd)           # means over LOCAL valid rays
    sdf_term = (e + f) * (nv_l / nv_g) * w_sdf
    total = total + torch.where(nv_l > 0, sdf_term, torch.zeros_like(sdf_term))
    total = total + w_eikonal * ret["gradient_error"].mean() * (nr_l / nr_g)
    glob = all_reduce_sum_(total.detach().clone().double(), group)
    return total, glob


class FlatGradReducer:
    """All-reduce(SUM) of the gradients of `params` through one persistent flat buffer."""

    def __init__(self, params, dtype=torch.float32):
        self.params = [p for p in params]
        self.sizes = [p.numel() for p in self.params]
        self.dtype = dtype
        self.flat = None

    def nbytes(self):
        return sum(self.sizes) * torch.empty((), dtype=self.dtype).element_size()

    def reduce(self, group=None):
        p0 = self.params[0]
        if self.flat is None or self.flat.device != p0.device:
            self.flat = torch.zeros(sum(self.sizes), dtype=self.dtype, device=p0.device)
        off = 0
        for p, k in zip(self.params, self.sizes):
            if p.grad is None:
                self.flat[off:off + k].zero_()
            else:
                self.flat[off:off + k].copy_(p.grad.reshape(-1))
            off += k
        all_reduce_sum_(self.flat, group)
        off = 0
        for p, k in zip(self.params, self.sizes):
            g = self.flat[off:off + k].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += k
