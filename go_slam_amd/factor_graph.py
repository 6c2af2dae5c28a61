"""Factor graph over keyframes: edge lists, per-edge GRU state and the `update` hot loop
(mirrors src/factor_graph.py:24-252 for the volume-correlation frontend path).

update() = reproject (HIP) -> motion features -> 4-level corr lookup (HIP, one launch) ->
UpdateModule (MIOpen convs) -> dense BA (HIP, no host round trips) -> convex upsampling.
"""
import torch

from .corr import AltCorrBlock, CorrBlock


def coords_grid(ht, wd, device):
    y, x = torch.meshgrid(torch.arange(ht, device=device).float(), torch.arange(wd, device=device).float(),
                          indexing="ij")
    return torch.stack([x, y], dim=-1)


class FactorGraph:
    def __init__(self, video, update_op, device="cuda:0", corr_impl="volume", max_factors=-1, upsample=False,
                 channels_last=True):
        """`channels_last`: keep the per-edge GRU state (net, inp) and the looked-up correlation
        features in NHWC memory so MIOpen runs its NHWC fp16 kernels (1.4x on the update operator
        on MI355X); shapes and values are unchanged."""
        assert corr_impl in ("volume", "alt")
        self.channels_last = channels_last
        self.video = video
        self.update_op = update_op
        self.device = torch.device(device)
        self.max_factors = max_factors
        self.corr_impl = corr_impl
        self.upsample = upsample
        self.ht, self.wd = video.ht, video.wd
        self.coords0 = coords_grid(self.ht, self.wd, self.device)
        z = lambda *s: torch.zeros(*s, device=self.device)
        self.ii = torch.zeros(0, dtype=torch.long, device=self.device)
        self.jj = torch.zeros(0, dtype=torch.long, device=self.device)
        self.age = torch.zeros(0, dtype=torch.long, device=self.device)
        self.corr, self.net, self.inp = None, None, None
        self.damping = 1e-6 * torch.ones_like(video.disps)
        self.target = z(1, 0, self.ht, self.wd, 2)
        self.weight = z(1, 0, self.ht, self.wd, 2)
        self.ii_inac = torch.zeros(0, dtype=torch.long, device=self.device)
        self.jj_inac = torch.zeros(0, dtype=torch.long, device=self.device)
        self.target_inac = z(1, 0, self.ht, self.wd, 2)
        self.weight_inac = z(1, 0, self.ht, self.wd, 2)

    def _edge_index(self, t0, t1, use_inactive):
        """Everything update() derives from the edge lists alone -- BA window, GraphAgg segments, damping
        rows, the active+inactive edge list -- computed once per graph topology on a host copy of ii/jj
        and cached until an edge list is replaced or written to.  The reference recomputes these every
        update with torch.unique / min / max / boolean indexing on the GPU (src/factor_graph.py:213-240),
        each a sort or a host sync; here a steady-state update issues none."""
        from .droid_net import build_segments
        tens = (self.ii, self.jj, self.ii_inac, self.jj_inac)
        key = tuple(x._version for x in tens) + (t0, t1, bool(use_inactive))
        c = getattr(self, "_eidx", None)
        if c is not None and c["key"] == key and all(a is b for a, b in zip(c["tens"], tens)):
            return c
        dev = self.device
        ii_c, jj_c = self.ii.cpu(), self.jj.cpu()
        a0 = max(1, int(ii_c.min()) + 1) if t0 is None else t0
        a0 = max(1, a0)
        a1 = (max(int(ii_c.max()), int(jj_c.max())) + 1) if t1 is None else t1
        seg = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in build_segments(ii_c).items()}
        c = {"key": key, "tens": tens, "t0": a0, "t1": a1, "seg": seg, "sel": None}
        ii_all = ii_c
        if use_inactive:
            iin, jin = self.ii_inac.cpu(), self.jj_inac.cpu()
            m = (iin >= a0 - 3) & (jin >= a0 - 3)
            c["sel"] = torch.nonzero(m).reshape(-1).to(dev)
            ii_all = torch.cat([iin[m], ii_c])
            c["ii"] = ii_all.to(dev)
            c["jj"] = torch.cat([jin[m], jj_c]).to(dev)
        else:
            c["ii"], c["jj"] = self.ii.contiguous(), self.jj.contiguous()
        c["damping_index"] = torch.unique(torch.cat([torch.arange(a0, a1), ii_all]), sorted=True).to(dev)
        self._eidx = c
        return c

    def _glue_fusable(self, coords1):
        return coords1.is_cuda and self.channels_last and self.target.dtype == torch.float32 \
            and self.target.is_contiguous() and coords1.is_contiguous()

    def _ba_buffers(self, idx, use_inactive):
        """[E_all,2,h,w] target / weight operands of droid_backends.ba, kept with the cached edge index: the
        inactive edges' rows are transposed once per edge set, the active rows are rewritten each update."""
        if "ba_target" not in idx:
            ht, wd = self.ht, self.wd
            E = self.ii.numel()
            if use_inactive and idx["sel"].numel():
                ti = self.target_inac[0, idx["sel"]].permute(0, 3, 1, 2)
                wi = self.weight_inac[0, idx["sel"]].permute(0, 3, 1, 2)
            else:
                ti = wi = torch.zeros(0, 2, ht, wd, device=self.device)
            n_in = ti.shape[0]
            bt = torch.empty(n_in + E, 2, ht, wd, device=self.device)
            bw = torch.empty(n_in + E, 2, ht, wd, device=self.device)
            bt[:n_in], bw[:n_in] = ti, wi
            idx["ba_target"], idx["ba_weight"], idx["n_inactive"] = bt, bw, n_in
        return idx["ba_target"], idx["ba_weight"], idx["n_inactive"]

    @torch.no_grad()
    def add_factors(self, ii, jj, remove=False):
        """add edges ii->jj (src/factor_graph.py:80-130): builds their correlation pyramids."""
        ii = torch.as_tensor(ii, dtype=torch.long, device=self.device).reshape(-1)
        jj = torch.as_tensor(jj, dtype=torch.long, device=self.device).reshape(-1)
        if ii.numel() == 0:
            return
        net = self._fmt(self.video.nets[ii]).unsqueeze(0)
        if self.corr_impl == "volume":              # the alt path correlates on the fly (no volumes)
            c = (ii == jj).long()                   # stereo edges read the right-view feature map
            fmap1 = self.video.fmaps[ii, 0].unsqueeze(0)
            fmap2 = self.video.fmaps[jj, c].unsqueeze(0)
            corr = CorrBlock(fmap1, fmap2, channels_last=self.channels_last)
            self.corr = corr if self.corr is None else self.corr.cat(corr)
            inp = self._fmt(self.video.inps[ii]).unsqueeze(0)
            self.inp = inp if self.inp is None else self._cat_edges(self.inp, inp)
        with torch.autocast("cuda", enabled=False):
            target, _ = self.video.reproject(ii, jj)
            weight = torch.zeros_like(target)
        self.ii = torch.cat([self.ii, ii])
        self.jj = torch.cat([self.jj, jj])
        self.age = torch.cat([self.age, torch.zeros_like(ii)])
        self.net = net if self.net is None else self._cat_edges(self.net, net)
        self.target = torch.cat([self.target, target], 1)
        self.weight = torch.cat([self.weight, weight], 1)

    def _fmt(self, x4):
        """[E,C,h,w] in the graph's memory format."""
        return x4.contiguous(memory_format=torch.channels_last) if self.channels_last else x4.contiguous()

    def _cat_edges(self, a5, b5):
        """cat two [1,E,C,h,w] state tensors over edges, keeping the 4-D memory format."""
        return self._fmt(torch.cat([a5[0], b5[0]], 0)).unsqueeze(0)

    @torch.no_grad()
    def rm_factors(self, mask, store=False):
        """drop edges (src/factor_graph.py:132-158), optionally keeping them as inactive."""
        if store:
            self.ii_inac = torch.cat([self.ii_inac, self.ii[mask]])
            self.jj_inac = torch.cat([self.jj_inac, self.jj[mask]])
            self.target_inac = torch.cat([self.target_inac, self.target[:, mask]], 1)
            self.weight_inac = torch.cat([self.weight_inac, self.weight[:, mask]], 1)
        keep = ~mask
        self.ii, self.jj, self.age = self.ii[keep], self.jj[keep], self.age[keep]
        if self.corr_impl == "volume":
            self.corr = self.corr[keep]
            self.inp = self._fmt(self.inp[0][keep]).unsqueeze(0)
        self.net = self._fmt(self.net[0][keep]).unsqueeze(0)
        self.target = self.target[:, keep]
        self.weight = self.weight[:, keep]

    @torch.no_grad()
    def update(self, t0=None, t1=None, iters=2, use_inactive=False, EPS=1e-7, motion_only=False):
        """run the update operator on the factor graph (src/factor_graph.py:199-252)."""
        idx = self._edge_index(t0, t1, use_inactive)
        t0, t1, seg = idx["t0"], idx["t1"], idx["seg"]
        coords1, mask = self.video.reproject(self.ii, self.jj)
        fused = self._glue_fusable(coords1)
        ht, wd = self.ht, self.wd
        E = self.ii.numel()
        if fused:
            from . import _lib
            L, st = _lib.lib(), _lib.stream_ptr(self.device)
            m4 = torch.empty(E, ht, wd, 4, dtype=torch.float16, device=self.device)
            _lib.check(L.gs_motion_features(_lib.ptr(coords1), _lib.ptr(self.target), _lib.ptr(m4), E, ht, wd, st),
                       "motion_features")
            motion = m4.permute(0, 3, 1, 2).unsqueeze(0)          # logical [1,E,4,h,w], NHWC memory
        else:
            motion = torch.cat([coords1 - self.coords0, self.target - coords1], dim=-1)
            motion = motion.permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)

        corr = self.corr(coords1)
        with torch.autocast("cuda", dtype=torch.float16):
            self.net, delta, weight, damping, upmask = self.update_op(
                self.net, self.inp, corr, motion, self.ii, self.jj, seg=seg)

        self.damping[seg["uniq"]] = damping.float()
        if fused and delta.dtype == torch.float32 and weight.dtype == torch.float32:
            # one pass: target = coords1 + delta, and the [E,2,h,w] BA operands written behind the
            # (cached, already transposed) inactive edges
            bt, bw, n_in = self._ba_buffers(idx, use_inactive)
            target_new = torch.empty_like(coords1)
            _lib.check(L.gs_ba_inputs(_lib.ptr(coords1), _lib.ptr(delta.contiguous()), _lib.ptr(weight.contiguous()),
                                      _lib.ptr(target_new), bt[n_in:].data_ptr(), bw[n_in:].data_ptr(), E, ht, wd, st),
                       "ba_inputs")
            self.target, self.weight = target_new, weight
            target, weight = bt, bw
        else:
            self.target = coords1 + delta.float()
            self.weight = weight.float()
            if use_inactive:
                sel = idx["sel"]
                target = torch.cat([self.target_inac[:, sel], self.target], 1)
                weight = torch.cat([self.weight_inac[:, sel], self.weight], 1)
            else:
                target, weight = self.target, self.weight
            target = target.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()
            weight = weight.view(-1, ht, wd, 2).permute(0, 3, 1, 2).contiguous()

        damping = 0.2 * self.damping[idx["damping_index"]].contiguous() + EPS

        self.video.ba(target, weight, damping, idx["ii"], idx["jj"], t0=t0, t1=t1, iters=iters,
                      lm=1e-4, ep=0.1, motion_only=motion_only)
        if self.upsample:
            self.video.upsample(seg["uniq"], upmask[0])
        self.age += 1

    @torch.no_grad()
    def update_lowmem(self, t0=None, t1=None, iters=2, use_inactive=False, EPS=1e-7, steps=8, max_t=None,
                      ba_type="dense", motion_only=False):
        """Reduced-memory update for global / loop-closure BA (src/factor_graph.py:255-321): alt-corr
        lookups in chunks of 13 source keyframes, one dense BA over all edges per step."""
        cur_t = self.video.counter
        t = max_t if max_t is not None else cur_t
        fm = self.video.fmaps[:cur_t + 2]
        num, rig, ch, ht, wd = fm.shape
        corr_op = AltCorrBlock(fm.reshape(1, num * rig, ch, ht, wd))
        if t0 is None:
            t0 = max(1, int(self.ii.min()) + 1)
        t0 = max(1, t0)
        if t1 is None:
            t1 = max(int(self.ii.max()), int(self.jj.max())) + 1
        for _ in range(steps):
            coords1, mask = self.video.reproject(self.ii, self.jj)
            motion = torch.cat([coords1 - self.coords0, self.target - coords1], dim=-1)
            motion = motion.permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
            s = 13
            lo, hi = int(self.ii.min()), int(self.ii.max())
            for i in range(lo, hi + 1, s):
                v = (self.ii >= i) & (self.ii < i + s)
                if int(v.sum()) < 1:
                    continue
                iis, jjs = self.ii[v], self.jj[v]
                corr1 = corr_op(coords1[:, v], rig * iis, rig * jjs + (iis == jjs).long())
                with torch.autocast("cuda", dtype=torch.float16):
                    net, delta, weight, damping, upmask = self.update_op(
                        self.net[:, v], self._fmt(self.video.inps[iis]).unsqueeze(0), corr1, motion[:, v], iis, jjs)
                    if self.upsample:
                        self.video.upsample(torch.unique(iis, sorted=True), upmask[0])
                self.net[:, v] = net.to(self.net.dtype)
                self.target[:, v] = coords1[:, v] + delta.float()
                self.weight[:, v] = weight.float()
                self.damping[torch.unique(iis, sorted=True)] = damping.float()
            damping_index = torch.unique(torch.cat([torch.arange(t0, t1, device=self.ii.device), self.ii]), sorted=True)
            damping = 0.2 * self.damping[damping_index].contiguous() + EPS
            target = self.target.view(-1, self.ht, self.wd, 2).permute(0, 3, 1, 2).contiguous()
            weight = self.weight.view(-1, self.ht, self.wd, 2).permute(0, 3, 1, 2).contiguous()
            lm, ep = (1e-4, 1e-1) if ba_type == "loop" else (1e-5, 1e-2)
            self.video.ba(target, weight, damping, self.ii.contiguous(), self.jj.contiguous(), t0=t0, t1=t1,
                          iters=iters, lm=lm, ep=ep, motion_only=motion_only, ba_type=ba_type)
            self.video.dirty[:t] = True
