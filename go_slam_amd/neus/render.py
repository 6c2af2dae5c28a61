"""Renderer.render_batch_ray / eval_points (src/render.py:29-175): same signatures; the sample
placement (far bound, stratified + near-surface samples, sort, dists -- ~30 ATen ops and a
torch.sort in the reference) is one HIP launch, `gs_render_sample`."""
import torch

from .. import _lib


class Renderer:
    def __init__(self, cfg=None, args=None, slam=None, points_batch_size=1e4, ray_batch_size=5e3,
                 N_samples=24, N_surface=48, perturb=1.0, lindisp=False, rand_pool_rows=1):
        self.ray_batch_size = int(ray_batch_size)
        self.points_batch_size = int(points_batch_size)
        r = (cfg or {}).get("rendering", {})
        self.lindisp = r.get("lindisp", lindisp)
        self.perturb = r.get("perturb", perturb)
        self.N_samples = r.get("N_samples", N_samples)
        self.N_surface = r.get("N_surface", N_surface)
        if self.lindisp:
            raise NotImplementedError("lindisp sampling is off in every reference config (configs/*.yaml)")
        # 1 (default): one `torch.rand(N_samples, device=...)` per batch -- the reference's own call (render.py:159), so a
        # seeded run consumes the device generator exactly as the reference does (the mapper's pixel draws between two
        # batches come from the same stream).  > 1: one [rows, N_samples] draw per `rows` batches (one launch less per
        # batch; NOT the same Philox consumption, so not the reference's numbers under a seed)
        self.rand_pool_rows = int(rand_pool_rows)
        self._lin = {}
        self._rand = {}

    def _linspace(self, steps, device):
        key = (steps, str(device))
        if key not in self._lin:
            self._lin[key] = torch.linspace(0, 1, steps=steps, device=device).float().contiguous()
        return self._lin[key]

    def _perturb_row(self, ns, device):
        """The `torch.rand(N_samples)` vector of one batch (render.py:159), shared by all its rays."""
        rows = self.rand_pool_rows
        if rows <= 1:
            return torch.rand(ns, device=device)
        key = (ns, str(device))
        pool, used = self._rand.get(key, (None, rows))
        if used >= rows:
            pool, used = torch.rand(rows, ns, device=device), 0
        self._rand[key] = (pool, used + 1)
        return pool[used]

    def sample(self, rays_o, rays_d, bound, gt_depth=None, perturb_rand=None, gt_max_dev=None):
        """z_vals, dists [N, N_samples + N_surface] (render.py:99-171).  `gt_max_dev` (device fp32[1], optional): the
        maximum depth measurement of the WHOLE batch when `gt_depth` is only one rank's shard of it -- the reference
        clamps every ray's far bound with `gt_depth.max()` of the batch (:121,:140), so a sharded step must not take
        the maximum over its own rays only."""
        dev = rays_o.device
        n = rays_o.shape[0]
        ns = self.N_samples
        nsurf = self.N_surface if gt_depth is not None else 0
        if n == 0:
            z = torch.zeros(0, ns + nsurf, dtype=torch.float32, device=dev)
            return z, z.clone()
        if gt_depth is not None:
            gt_depth = gt_depth.reshape(-1).float().contiguous()
            gt_max = 0.0
            if gt_max_dev is None:
                # the batch maximum never leaves the device (the reference's .max() at :121,:140 costs a host sync per
                # batch): taken inside the sampling launch where that is possible (-inf asks for it), by a reduction
                # launch in front of it otherwise
                if ns <= 64 and nsurf <= 64 and n <= 65536 and gt_depth.data_ptr() % 16 == 0:
                    gt_max = float("-inf")
                else:
                    gt_max_dev = gt_depth.max().reshape(1)
        else:
            gt_max, gt_max_dev = 0.0, None
        if self.perturb > 0 and perturb_rand is None:
            perturb_rand = self._perturb_row(ns, dev)           # one vector shared by all rays (:159)
        z = torch.empty(n, ns + nsurf, dtype=torch.float32, device=dev)
        d = torch.empty_like(z)
        with torch.cuda.device(dev):
            rc = _lib.lib().gs_render_sample(
                _lib.ptr(rays_o.detach().float().contiguous()), _lib.ptr(rays_d.detach().float().contiguous()),
                _lib.ptr(gt_depth), _lib.ptr(bound.to(dev).float().contiguous()), _lib.ptr(self._linspace(ns, dev)),
                _lib.ptr(self._linspace(nsurf, dev) if nsurf else None),
                _lib.ptr(perturb_rand.float().contiguous() if perturb_rand is not None else None), gt_max,
                _lib.ptr(gt_max_dev), _lib.ptr(z), _lib.ptr(d), n, ns, nsurf, _lib.stream_ptr(dev))
        _lib.check(rc, "Renderer.sample")
        return z, d

    def eval_points(self, rays_o, rays_d, z_vals, dists, net, render_params):
        """src/render.py:29-71: chunks of `points_batch_size` rays."""
        out = {}
        for ro, rd, zv, ds in zip(torch.split(rays_o, self.points_batch_size), torch.split(rays_d, self.points_batch_size),
                                  torch.split(z_vals, self.points_batch_size), torch.split(dists, self.points_batch_size)):
            o = net(ro, rd, zv, ds, render_params=render_params)
            if not out:
                out = o
                continue
            for k, v in o.items():
                out[k] = torch.cat([out[k], v], dim=0) if torch.is_tensor(v) else v
        return out

    def render_batch_ray(self, rays_o, rays_d, net, render_params=None, device="cuda:0", gt_depth=None):
        z_vals, dists = self.sample(rays_o, rays_d, net.bound, gt_depth)
        return self.eval_points(rays_o, rays_d, z_vals, dists, net, render_params)
