"""Factor graph over keyframes: edge lists, per-edge GRU state and the `update` hot loop
(mirrors src/factor_graph.py:24-252 for the volume-correlation frontend path).

update() = reproject (HIP) -> motion features -> 4-level corr lookup (HIP, one launch) ->
UpdateModule (gs_conv3x3 / MIOpen convolutions + HIP epilogues) -> dense BA (HIP, no host round trips) -> convex
upsampling.
"""
import os

import numpy as np
import torch

from .corr import AltCorrBlock, CorrBlock, CorrPool

# FactorGraph.update: keep the BA's index tables per edge set (GS_BA_REUSE_TABLES); 0 = rebuild them in every call
BA_TABLES = os.environ.get("GOSLAM_BA_TABLES", "1") == "1"


def coords_grid(ht, wd, device):
    y, x = torch.meshgrid(torch.arange(ht, device=device).float(), torch.arange(wd, device=device).float(),
                          indexing="ij")
    return torch.stack([x, y], dim=-1)


def _segments_np(index):
    """droid_net.build_segments on a host numpy index (the same five entries: uniq / ix int64, offsets / order int32, n).
    The edge tables are a few hundred integers; as torch CPU operators every step of their construction costs 5-10 us of
    dispatch -- 0.27 ms per edge index, 0.79 ms per loop-closure chunk index, all of it with the GPU idle behind the
    device-to-host read that feeds it -- where numpy takes ~1 us."""
    uniq, ix = np.unique(index, return_inverse=True)
    ix = ix.reshape(-1).astype(np.int64)
    order = np.argsort(ix, kind="stable").astype(np.int32)
    offsets = np.zeros(uniq.size + 1, dtype=np.int32)
    offsets[1:] = np.cumsum(np.bincount(ix, minlength=uniq.size))
    return {"uniq": uniq.astype(np.int64), "ix": ix, "offsets": offsets, "order": order, "n": int(uniq.size)}


def _as_tensors(d):
    return {k: (torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v) for k, v in d.items()}


def upload_tables(host, device):
    """{name: host tensor | other} -> the same dict with every tensor on `device`, through ONE host-to-device copy per dtype
    (the pieces are views into one buffer, each starting on a 16-byte boundary).  A `.to(device)` of a pageable host tensor
    is a blocking copy enqueued behind everything the stream holds: nine of them per edge-set change (the edge index) and
    six per 13-keyframe chunk (the loop-closure BA's chunk index) were that many pipeline drains per keyframe."""
    out = dict(host)
    if torch.device(device).type == "cpu":
        return out
    if os.environ.get("GOSLAM_BATCH_UPLOADS", "1") == "0":      # (A/B runs: one blocking copy per table, the round-5 form)
        return {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in host.items()}
    groups = {}
    for k, v in host.items():
        if torch.is_tensor(v):
            groups.setdefault(v.dtype, []).append(k)
    for dt, keys in groups.items():
        per16 = max(1, 16 // torch.empty(0, dtype=dt).element_size())
        pieces, spans, off = [], [], 0
        for k in keys:
            n = host[k].numel()
            pad = (-n) % per16
            pieces.append(host[k].reshape(-1))
            if pad:
                pieces.append(torch.zeros(pad, dtype=dt))
            spans.append((k, off, n))
            off += n + pad
        flat = torch.cat(pieces).to(device) if off else torch.zeros(0, dtype=dt, device=device)
        for k, o, n in spans:
            out[k] = flat[o:o + n].view(host[k].shape)
    return out


class FactorGraph:
    def __init__(self, video, update_op, device="cuda:0", corr_impl="volume", max_factors=-1, upsample=False,
                 channels_last=True):
        """`channels_last`: keep the per-edge GRU state (net, inp) and the looked-up correlation
        features in NHWC memory so MIOpen runs its NHWC fp16 kernels (1.4x on the update operator
        on MI355X); shapes and values are unchanged."""
        assert corr_impl in ("volume", "alt")
        self.channels_last = channels_last
        self.pool_volumes = True        # False: CorrBlock with the reference's cat / [keep] copies (A/B, tests)
        self.video = video
        self.update_op = update_op
        self.device = torch.device(device)
        self.max_factors = max_factors
        self.corr_impl = corr_impl
        self.upsample = upsample
        # 1/8-resolution map size, read off the buffers: unambiguous for this package's DepthVideo and for a
        # reference-style video object alike (whose .ht is the full-resolution height, src/factor_graph.py:19-20)
        self.ht, self.wd = (int(x) for x in video.disps.shape[-2:])
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
        self.ii_bad = torch.zeros(0, dtype=torch.long, device=self.device)
        self.jj_bad = torch.zeros(0, dtype=torch.long, device=self.device)
        self.target_inac = z(1, 0, self.ht, self.wd, 2)
        self.weight_inac = z(1, 0, self.ht, self.wd, 2)

    def _edge_index(self, t0, t1, use_inactive):
        """Everything update() derives from the edge lists alone -- BA window, GraphAgg segments, damping
        rows, the active+inactive edge list -- computed once per graph topology on a host copy of ii/jj
        and cached until an edge list is replaced or written to.  The reference recomputes these every
        update with torch.unique / min / max / boolean indexing on the GPU (src/factor_graph.py:213-240),
        each a sort or a host sync; here a steady-state update issues none."""
        tens = (self.ii, self.jj, self.ii_inac, self.jj_inac)
        key = tuple(x._version for x in tens) + (t0, t1, bool(use_inactive))
        c = getattr(self, "_eidx", None)
        if c is not None and c["key"] == key and all(a is b for a, b in zip(c["tens"], tens)):
            return c
        dev = self.device
        E, Ei = self.ii.numel(), self.ii_inac.numel()
        host = torch.cat([self.ii, self.jj, self.ii_inac, self.jj_inac]).cpu().numpy()     # ONE device-to-host read per edge set
        ii_c, jj_c = host[:E], host[E:2 * E]
        a0 = max(1, int(ii_c.min()) + 1) if t0 is None else t0
        a0 = max(1, a0)
        a1 = (max(int(ii_c.max()), int(jj_c.max())) + 1) if t1 is None else t1
        seg_h = _segments_np(ii_c)                       # (numpy on the host copy: see _segments_np)
        c = {"key": key, "tens": tens, "t0": a0, "t1": a1, "seg": None, "sel": None, "ii_min": int(ii_c.min())}
        up = {"seg." + k: v for k, v in seg_h.items() if isinstance(v, np.ndarray)}   # everything below goes up in ONE copy per dtype
        ii_all = ii_c
        if use_inactive:
            iin, jin = host[2 * E:2 * E + Ei], host[2 * E + Ei:]
            m = (iin >= a0 - 3) & (jin >= a0 - 3)
            up["sel"] = np.nonzero(m)[0].astype(np.int64)
            ii_all = np.concatenate([iin[m], ii_c])
            up["ii"] = ii_all
            up["jj"] = np.concatenate([jin[m], jj_c])
        else:
            c["ii"], c["jj"] = self.ii.contiguous(), self.jj.contiguous()
        dindex = np.unique(np.concatenate([np.arange(a0, a1, dtype=np.int64), ii_all]))
        up["damping_index"] = dindex
        # row of the operator's eta output (one per unique source keyframe of the ACTIVE edges) for every damping row
        uniq = seg_h["uniq"]
        if uniq.size:
            pos = np.minimum(np.searchsorted(uniq, dindex), uniq.size - 1)
            inv = np.where(uniq[pos] == dindex, pos, -1)
        else:
            inv = np.full(dindex.shape, -1)
        up["damping_inv"] = inv.astype(np.int32)
        up = upload_tables(_as_tensors(up), dev)
        c["seg"] = {k: (up["seg." + k] if isinstance(v, np.ndarray) else v) for k, v in seg_h.items()}
        for k in ("sel", "ii", "jj", "damping_index", "damping_inv"):
            if k in up:
                c[k] = up[k]
        uniq = torch.from_numpy(uniq)
        c["uniq_host"] = uniq
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
        ii, jj = self._filter_repeated_edges(ii, jj)
        if ii.numel() == 0:
            return
        # limit on the number of factors (src/factor_graph.py:99-103): the oldest edges become inactive.  The
        # reference's mask is `argsort(age) >= max_factors - n_new` (the permutation itself, not the rank) --
        # kept literally so the same edges retire.
        if self.max_factors > 0 and self.ii.numel() + ii.numel() > self.max_factors and self.corr is not None \
                and remove:
            perm = torch.argsort(self.age, descending=False)
            self.rm_factors(perm >= self.max_factors - ii.numel(), store=True)
        net = self._fmt(self.video.nets[ii]).unsqueeze(0)
        if self.corr_impl == "volume":              # the alt path correlates on the fly (no volumes)
            c = (ii == jj).long()                   # stereo edges read the right-view feature map
            fmap1 = self.video.fmaps[ii, 0].unsqueeze(0)
            fmap2 = self.video.fmaps[jj, c].unsqueeze(0)
            if self.corr is None and self.pool_volumes and CorrPool.supported(fmap1, self.channels_last):
                # the edges' volumes in slots of a capacity buffer: adding / dropping edges moves none (corr.CorrPool)
                self.corr = CorrPool(self.ht, self.wd, self.device, capacity=max(self.max_factors, 48) + 32)
            if isinstance(self.corr, CorrPool):
                self.corr.append(fmap1, fmap2)
            else:
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
        """drop edges (src/factor_graph.py:132-158), optionally keeping them as inactive.  The reference indexes nine
        tensors with the boolean mask -- a `nonzero` (four scan launches + a host sync) each; here the mask becomes two
        index lists once and everything is an index_select."""
        mask = mask.to(device=self.device, dtype=torch.bool).reshape(-1)
        if mask.is_cuda:                                # both lists from ONE device-to-host read
            order = torch.argsort(mask.to(torch.uint8), stable=True)     # kept edges first, in order; then the dropped
            n_keep = mask.numel() - int(mask.sum())
            keep, gone = order[:n_keep], order[n_keep:]
        else:
            keep, gone = torch.nonzero(~mask).reshape(-1), torch.nonzero(mask).reshape(-1)
        if gone.numel() == 0:
            return
        if store:
            self.ii_inac = torch.cat([self.ii_inac, self.ii.index_select(0, gone)])
            self.jj_inac = torch.cat([self.jj_inac, self.jj.index_select(0, gone)])
            self.target_inac = torch.cat([self.target_inac, self.target.index_select(1, gone)], 1)
            self.weight_inac = torch.cat([self.weight_inac, self.weight.index_select(1, gone)], 1)
        self.ii, self.jj, self.age = (x.index_select(0, keep) for x in (self.ii, self.jj, self.age))
        if self.corr_impl == "volume" and self.corr is not None:
            self.corr = self.corr[keep]
        if self.inp is not None:
            self.inp = self._select_edges(self.inp, keep)
        if self.net is not None:
            self.net = self._select_edges(self.net, keep)
        self.target = self.target.index_select(1, keep)
        self.weight = self.weight.index_select(1, keep)

    # ---- edge-set management (src/factor_graph.py:43-53, 70-83, 159-197, 368-450) -----------------------
    @staticmethod
    def _edges_on_host(*pairs):
        """(i, j) tuples of the given (ii, jj) tensor pairs: one device-to-host copy per tensor (the
        reference calls .item() per element, a host sync each)."""
        out = []
        for a, b in pairs:
            out += list(zip(a.cpu().tolist(), b.cpu().tolist()))
        return out

    def _filter_repeated_edges(self, ii, jj):
        """drop proposed edges that are already active or inactive (src/factor_graph.py:43-53).  As in the
        reference, duplicates WITHIN the proposal are kept."""
        if ii.numel() == 0 or self.ii.numel() + self.ii_inac.numel() == 0:
            return ii, jj
        if ii.is_cuda:                                  # one key per edge, membership on the device, ONE host read
            have = torch.cat([self.ii, self.ii_inac]) * (1 << 20) + torch.cat([self.jj, self.jj_inac])
            keep = torch.nonzero(~torch.isin(ii * (1 << 20) + jj, have)).reshape(-1)
            if keep.numel() == ii.numel():
                return ii, jj
            return ii.index_select(0, keep), jj.index_select(0, keep)
        have = set(self._edges_on_host((self.ii, self.jj), (self.ii_inac, self.jj_inac)))
        keep = torch.tensor([e not in have for e in zip(ii.cpu().tolist(), jj.cpu().tolist())],
                            dtype=torch.bool, device=ii.device)
        return ii[keep], jj[keep]

    @torch.no_grad()
    def filter_edges(self):
        """remove low-confidence long-range edges and remember them as bad (src/factor_graph.py:70-77)."""
        conf = self.weight.mean(dim=(0, 2, 3, 4))
        mask = ((self.ii - self.jj).abs() > 2) & (conf < 1e-3)
        self.ii_bad = torch.cat([self.ii_bad, self.ii[mask]])
        self.jj_bad = torch.cat([self.jj_bad, self.jj[mask]])
        self.rm_factors(mask, store=False)

    @torch.no_grad()
    def clear_edges(self):
        """src/factor_graph.py:79-82"""
        if self.ii.numel():                     # every edge goes: nothing to read back (the reference's mask form costs a
            if self.corr_impl == "volume" and self.corr is not None:                 # reduction and a host sync)
                self.corr = self.corr[torch.zeros(0, dtype=torch.long, device=self.ii.device)]
            self.ii, self.jj, self.age = (x.new_empty((0,)) for x in (self.ii, self.jj, self.age))
            self.target, self.weight = (x.new_empty((x.shape[0], 0) + tuple(x.shape[2:])) for x in (self.target, self.weight))
        self.net = None
        self.inp = None

    # per-keyframe buffers the reference shifts in rm_keyframe (src/factor_graph.py:161-179); the host
    # mirror of DepthVideo holds a subset, a drop-in caller's video object may hold all of them
    _KEYFRAME_BUFFERS = ("timestamp", "images", "dirty", "red", "poses", "poses_gt", "disps", "disps_sens",
                         "disps_up", "depths_gt", "intrinsics", "poses_filtered", "disps_filtered",
                         "mask_filtered", "update_priority", "nets", "inps", "fmaps")

    @torch.no_grad()
    def rm_keyframe(self, ix):
        """drop keyframe `ix`: slot ix+1 moves down and every edge touching ix goes, later indices shift by
        one (src/factor_graph.py:159-197).  The edge lists are REPLACED rather than edited in place, which is
        what invalidates the cached edge index."""
        v = self.video
        for name in self._KEYFRAME_BUFFERS:
            buf = getattr(v, name, None)
            if torch.is_tensor(buf):
                buf[ix] = buf[ix + 1]
        m = (self.ii_inac == ix) | (self.jj_inac == ix)
        self.ii_inac = self.ii_inac - (self.ii_inac >= ix).long()
        self.jj_inac = self.jj_inac - (self.jj_inac >= ix).long()
        if bool(m.any()):
            keep = ~m
            self.ii_inac, self.jj_inac = self.ii_inac[keep], self.jj_inac[keep]
            self.target_inac, self.weight_inac = self.target_inac[:, keep], self.weight_inac[:, keep]
        m = (self.ii == ix) | (self.jj == ix)
        self.ii = self.ii - (self.ii >= ix).long()
        self.jj = self.jj - (self.jj >= ix).long()
        self.rm_factors(m, store=False)

    def add_neighborhood_factors(self, t0, t1, r=3):
        """edges between all frames of [t0, t1) at most r apart (src/factor_graph.py:368-381)."""
        ii, jj = torch.meshgrid(torch.arange(t0, t1), torch.arange(t0, t1), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        c = 1 if getattr(self.video, "stereo", False) else 0
        keep = ((ii - jj).abs() > c) & ((ii - jj).abs() <= r)
        self.add_factors(ii[keep].to(self.device), jj[keep].to(self.device))

    @staticmethod
    def propose_proximity_edges(d, existing, t0, t1, t, rad, nms, thresh, max_factors, stereo):
        """Greedy edge proposal with non-maximum suppression (src/factor_graph.py:402-447), on the host.

        d: float numpy array [t - t0, t - t1] of frame distances, inf where the pair is not a candidate
        (modified in place); existing: (i, j) pairs already active / bad / inactive.  Returns the list of
        proposed (i, j) pairs in the reference's order.  The index expressions are the reference's own;
        NumPy resolves negative indices and slice bounds exactly as torch does, so the corner cases
        (j < t1, windows clipped at the border) come out the same."""
        import numpy as np
        ilen, jlen = t - t0, t - t1

        def suppress(di, dj):
            d[max(0, di - nms):min(ilen, di + nms + 1), max(0, dj - nms):min(jlen, dj + nms + 1)] = np.inf

        for i, j in existing:                         # edges built before
            if t0 <= i < t and t1 <= j < t:
                d[i - t0, j - t1] = np.inf
                suppress(i - t0, j - t1)
        es = []
        for i in range(t0, t):                        # local window [i - rad, i)
            if stereo:
                es.append((i, i))
                d[i - t0, i - t1] = np.inf
            for j in range(max(i - rad, 0), i):
                es += [(i, j), (j, i)]
                d[i - t0, j - t1] = np.inf
                suppress(i - t0, j - t1)
        flat = d.reshape(-1)
        order = np.argsort(flat, kind="stable")       # distance from small to big
        order = order[flat[order] <= thresh]
        for k in order.tolist():
            di, dj = k // jlen, k % jlen
            if d[di, dj] > thresh:                    # suppressed by an earlier pick
                continue
            if len(es) > max_factors:
                break
            i, j = t0 + di, t1 + dj
            es += [(i, j), (j, i)]                    # bidirectional
            suppress(di, dj)
        return es

    @staticmethod
    def propose_edges_on_device(raw, existing, i0, j0, t, rad, nms, cut, thresh, max_factors, stereo, jmin, loop):
        """The same greedy proposal on the GPU (csrc/edge_nms.hip: gs_edge_prep -> stable device sort -> gs_edge_greedy):
        `raw` f32 [(t - i0) * (t - j0)] stays in HBM, `existing` = (ii, jj) device tensors or None.  ONE int comes back
        to the host (the edge count sizes the edge tensors).  Returns an int64 [n, 2] device tensor in the reference's
        order -- the order and content of propose_proximity_edges / backend.propose_backend_edges (GPU test)."""
        from . import _lib
        dev = raw.device
        n_local = sum((1 if stereo else 0) + 2 * max(i - max(i - rad, jmin), 0) for i in range(i0, t))
        cap = n_local + max(int(max_factors), 0) + 16
        raw = raw.detach().float().contiguous()
        d_work = torch.empty_like(raw)
        es = torch.zeros(cap, 2, dtype=torch.long, device=dev)
        count = torch.zeros(1, dtype=torch.int32, device=dev)
        ex_i, ex_j = (None, None) if existing is None else (existing[0].long().contiguous(), existing[1].long().contiguous())
        n_ex = 0 if existing is None else int(ex_i.numel())
        L, st = _lib.lib(), _lib.stream_ptr(dev)
        with torch.cuda.device(dev):
            _lib.check(L.gs_edge_prep(_lib.ptr(raw), _lib.ptr(d_work), _lib.ptr(ex_i), _lib.ptr(ex_j), n_ex, _lib.ptr(es),
                                      _lib.ptr(count), cap, i0, j0, t, rad, nms, float(cut), int(bool(stereo)), jmin, st),
                       "edge_prep")
            vals, order = torch.sort(d_work, stable=True)
            _lib.check(L.gs_edge_greedy(_lib.ptr(raw), _lib.ptr(vals), _lib.ptr(order), _lib.ptr(es), _lib.ptr(count), cap,
                                        i0, j0, t, nms, float(thresh), int(max_factors), int(bool(loop)), st),
                       "edge_greedy")
        return es[:int(count)]

    @torch.no_grad()
    def add_proximity_factors(self, t0=0, t1=0, rad=2, nms=2, beta=0.25, thresh=16.0, remove=False, max_t=None):
        """add edges based on frame distance (src/factor_graph.py:383-450): one frame_distance launch pair, then the
        greedy NMS -- on the GPU for a CUDA video (the distance matrix stays in HBM, one int is read back); on the CPU
        one copy of the matrix and a NumPy loop (the reference keeps d on the GPU and syncs once per candidate)."""
        cnt = self.video.counter
        t = max_t if max_t is not None else int(getattr(cnt, "value", cnt))
        if t <= t0 or t <= t1:
            return
        vdev = torch.device(self.device)         # (index grids built where they are used: a host grid is two blocking uploads)
        ii, jj = torch.meshgrid(torch.arange(t0, t, device=vdev), torch.arange(t1, t, device=vdev), indexing="ij")
        ii, jj = ii.reshape(-1), jj.reshape(-1)
        d_dev = self.video.distance(ii, jj, beta=beta)
        if d_dev.is_cuda and (t - t0) * (t - t1) <= 512 * 512:
            existing = (torch.cat([self.ii, self.ii_bad, self.ii_inac]), torch.cat([self.jj, self.jj_bad, self.jj_inac]))
            e = FactorGraph.propose_edges_on_device(d_dev, existing, t0, t1, t, rad, nms, 100.0, thresh, self.max_factors,
                                                    bool(getattr(self.video, "stereo", False)), 0, False)
            if e.shape[0]:
                self.add_factors(e[:, 0].contiguous(), e[:, 1].contiguous(), remove)
            return
        d = d_dev.detach().float().cpu()
        ii, jj = ii.cpu(), jj.cpu()
        d[ii - rad < jj] = float("inf")
        d[d > 100] = float("inf")
        d = d.reshape(t - t0, t - t1).numpy().copy()
        existing = self._edges_on_host((self.ii, self.jj), (self.ii_bad, self.jj_bad), (self.ii_inac, self.jj_inac))
        es = FactorGraph.propose_proximity_edges(d, existing, t0, t1, t, rad, nms, thresh, self.max_factors,
                                                 bool(getattr(self.video, "stereo", False)))
        if not es:
            return
        e = torch.tensor(es, dtype=torch.long, device=self.device)
        self.add_factors(e[:, 0], e[:, 1], remove)

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

        # (a deferred lookup: the operator's fast path evaluates it fused with corr_encoder[0], anything else materialises it)
        corr = self.corr.lazy(coords1) if hasattr(self.corr, "lazy") else self.corr(coords1)
        with torch.autocast("cuda", dtype=torch.float16):
            self.net, delta, weight, damping, upmask = self.update_op(
                self.net, self.inp, corr, motion, self.ii, self.jj, seg=seg)

        fuse_damping = fused and damping.dtype == torch.float32 and damping.is_contiguous() and \
            self.damping.is_contiguous() and damping.shape[1] == idx["uniq_host"].numel()
        if not fuse_damping:
            self.damping[seg["uniq"]] = damping.float()
        eta_rows = damping
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

        if fuse_damping:
            # self.damping[uniq] = eta and 0.2 * self.damping[index] + EPS in one launch
            damping = torch.empty(idx["damping_index"].numel(), ht, wd, device=self.device)
            _lib.check(L.gs_damping_rows(_lib.ptr(eta_rows), _lib.ptr(idx["damping_inv"]), _lib.ptr(idx["damping_index"]),
                                         _lib.ptr(self.damping), _lib.ptr(damping), damping.shape[0], ht * wd, 0.2,
                                         float(EPS), st), "damping_rows")
        else:
            damping = 0.2 * self.damping[idx["damping_index"]].contiguous() + EPS

        # (the cached edge index also keeps the BA's index tables: built by the first call on this edge set, reused after)
        tables = idx.setdefault("ba_tables", {}) if (BA_TABLES and getattr(self.video, "ba_accepts_tables", False)) else None
        if tables is None:
            self.video.ba(target, weight, damping, idx["ii"], idx["jj"], t0=t0, t1=t1, iters=iters,
                          lm=1e-4, ep=0.1, motion_only=motion_only)
        else:
            self.video.ba(target, weight, damping, idx["ii"], idx["jj"], t0=t0, t1=t1, iters=iters,
                          lm=1e-4, ep=0.1, motion_only=motion_only, tables=tables)
        if self.upsample:
            self.video.upsample(seg["uniq"], upmask[0])
        self.age += 1

    def _select_edges(self, x5, sel):
        """x5[:, sel] for a [1,E,C,h,w] state tensor, keeping the graph's memory format (index_select on the NHWC view:
        rows of h*w*C contiguous halves, no layout conversion afterwards)."""
        x4 = x5[0]
        if self.channels_last and x4.is_contiguous(memory_format=torch.channels_last):
            return x4.permute(0, 2, 3, 1).index_select(0, sel).permute(0, 3, 1, 2).unsqueeze(0)
        return x4.index_select(0, sel).unsqueeze(0)

    def _lowmem_index(self, t0, t1, rig):
        """What update_lowmem derives from the edge lists alone, computed once per topology on a host copy and cached
        (the reference recomputes it per step and per chunk with boolean masks, `.sum()` / `.min()` / `.max()` reads
        and torch.unique -- two to four host syncs per 13-keyframe chunk, src/factor_graph.py:266-299): the BA window,
        the damping rows, and per chunk the edge selection, the alt-corr pyramid indices (`rig`-strided, right view for
        stereo pairs), the source keyframes to upsample and the GraphAgg segments."""
        tens = (self.ii, self.jj)
        # (the chunks keep laid-out copies of video.inps rows: a torch write to that buffer -- a keyframe appended or
        # shifted by rm_keyframe -- rebuilds them; writers torch cannot see use DepthVideo.add_write_hook)
        key = tuple(x._version for x in tens) + (t0, t1, rig, getattr(getattr(self.video, "inps", None), "_version", 0))
        c = getattr(self, "_lidx", None)
        if c is not None and c["key"] == key and all(a is b for a, b in zip(c["tens"], tens)):
            return c
        dev = self.device
        host = torch.stack([self.ii, self.jj]).cpu().numpy()       # ONE device-to-host read; the tables below in numpy
        ii_c, jj_c = host[0], host[1]                               # (see _segments_np)
        a0 = max(1, int(ii_c.min()) + 1) if t0 is None else t0
        a0 = max(1, a0)
        a1 = (max(int(ii_c.max()), int(jj_c.max())) + 1) if t1 is None else t1
        chunks, up, segs = [], {}, []
        s = 13
        for i in range(int(ii_c.min()), int(ii_c.max()) + 1, s):
            sel = np.nonzero((ii_c >= i) & (ii_c < i + s))[0].astype(np.int64)
            if sel.size < 1:
                continue
            iis, jjs = ii_c[sel], jj_c[sel]
            n = len(segs)
            up.update({f"{n}.sel": sel, f"{n}.ii": iis, f"{n}.jj": jjs, f"{n}.corr_ii": rig * iis,
                       f"{n}.corr_jj": rig * jjs + (iis == jjs).astype(np.int64), f"{n}.uniq": np.unique(iis)})
            segs.append(_segments_np(iis))
            up.update({f"{n}.seg.{k}": v for k, v in segs[-1].items() if isinstance(v, np.ndarray)})
        up["damping_index"] = np.unique(np.concatenate([np.arange(a0, a1, dtype=np.int64), ii_c]))
        up = _as_tensors(up)
        up = upload_tables(up, dev)                     # every chunk's tables in ONE copy per dtype
        for n, seg_h in enumerate(segs):
            ck = {k: up[f"{n}.{k}"] for k in ("sel", "ii", "jj", "corr_ii", "corr_jj", "uniq")}
            seg = {k: (up[f"{n}.seg.{k}"] if isinstance(v, np.ndarray) else v) for k, v in seg_h.items()}
            ck["seg_kw"] = {"seg": seg} if getattr(self.update_op, "_forward_fast", None) is not None else {}
            # (the chunk's context features are per-keyframe constants: gathered and laid out once per edge set -- the
            # index cache is rebuilt whenever an edge list or, through rm_keyframe, a keyframe slot changes -- instead of
            # once per step and chunk; 23 MB per chunk at 30 x 40)
            # (dict.setdefault evaluates its default on EVERY call: until round 5's last day this "cache" redid the gather
            # and the layout copy per chunk and step and threw the result away -- 16 x 42 us of the stress step)
            def _inp(ix=ck["ii"], box=ck):
                v = box.get("inp_cached")
                if v is None:
                    v = box["inp_cached"] = self._fmt(self.video.inps[ix]).unsqueeze(0)
                return v
            ck["inp"] = _inp
            chunks.append(ck)
        c = {"key": key, "tens": tens, "t0": a0, "t1": a1, "chunks": chunks, "ii": self.ii.contiguous(),
             "jj": self.jj.contiguous(), "damping_index": up["damping_index"]}
        self._lidx = c
        return c

    @torch.no_grad()
    def update_lowmem(self, t0=None, t1=None, iters=2, use_inactive=False, EPS=1e-7, steps=8, max_t=None,
                      ba_type="dense", motion_only=False):
        """Reduced-memory update for global / loop-closure BA (src/factor_graph.py:255-321): alt-corr
        lookups in chunks of 13 source keyframes, one dense BA over all edges per step."""
        cur_t = self.video.counter
        cur_t = int(getattr(cur_t, "value", cur_t))     # the reference's counter is a multiprocessing.Value
        t = max_t if max_t is not None else cur_t
        fm = self.video.fmaps[:cur_t + 2]
        num, rig, ch, ht, wd = fm.shape
        corr_op = AltCorrBlock(fm.reshape(1, num * rig, ch, ht, wd))
        lookup = getattr(corr_op, "lookup", corr_op)     # (all four levels in one launch, fp16 channels-last features)
        idx = self._lowmem_index(t0, t1, rig)
        t0, t1 = idx["t0"], idx["t1"]
        ii_all, jj_all = idx["ii"], idx["jj"]
        cl = torch.channels_last
        for _ in range(steps):
            coords1, mask = self.video.reproject(ii_all, jj_all)
            # the chunk glue as ONE gather and ONE scatter launch (csrc/lowmem_glue.hip) when the state is in the layouts the
            # kernels read: fp32 contiguous coordinates / targets / weights, fp16 NHWC recurrent state
            fast = (coords1.is_cuda and self.channels_last and coords1.dtype == torch.float32 and coords1.is_contiguous()
                    and self.target.dtype == torch.float32 and self.target.is_contiguous()
                    and self.weight.dtype == torch.float32 and self.weight.is_contiguous()
                    and self.net.dtype == torch.float16 and self.net[0].is_contiguous(memory_format=cl)
                    and self.net.shape[2] == 128          # gs_lowmem_gather / _scatter move 16 uint4 of state per pixel
                    and tuple(self.target.shape) == tuple(coords1.shape) == tuple(self.weight.shape))
            if fast:
                from . import _lib
                L, st = _lib.lib(), _lib.stream_ptr(self.device)
                ht, wd = self.ht, self.wd
            else:
                motion = torch.cat([coords1 - self.coords0, self.target - coords1], dim=-1)
                motion = motion.permute(0, 1, 4, 2, 3).clamp(-64.0, 64.0)
            for ck in idx["chunks"]:                    # 13 source keyframes at a time
                sel, iis, jjs = ck["sel"], ck["ii"], ck["jj"]
                if fast:
                    n = sel.numel()
                    c1 = torch.empty(1, n, ht, wd, 2, dtype=torch.float32, device=self.device)
                    m4 = torch.empty(n, ht, wd, 4, dtype=torch.float16, device=self.device)
                    net_c = torch.empty(n, ht, wd, 128, dtype=torch.float16, device=self.device)
                    _lib.check(L.gs_lowmem_gather(_lib.ptr(coords1), _lib.ptr(self.target), _lib.ptr(self.net),
                                                  _lib.ptr(sel), _lib.ptr(c1), _lib.ptr(m4), _lib.ptr(net_c), n, ht, wd, st),
                               "lowmem_gather")
                    net_in = net_c.permute(0, 3, 1, 2).unsqueeze(0)          # logical [1,n,128,h,w], NHWC memory
                    motion_c = m4.permute(0, 3, 1, 2).unsqueeze(0)           # logical [1,n,4,h,w], NHWC memory
                else:
                    c1 = coords1.index_select(1, sel)
                    net_in, motion_c = self._select_edges(self.net, sel), motion.index_select(1, sel)
                corr1 = lookup(c1, ck["corr_ii"], ck["corr_jj"])
                with torch.autocast("cuda", dtype=torch.float16):
                    net, delta, weight, damping, upmask = self.update_op(
                        net_in, ck["inp"](), corr1, motion_c, iis, jjs, **ck["seg_kw"])
                    if self.upsample:
                        self.video.upsample(ck["uniq"], upmask[0])
                if fast and net.dtype == torch.float16 and net[0].is_contiguous(memory_format=cl) \
                        and tuple(net.shape) == (1, sel.numel(), 128, ht, wd) \
                        and delta.dtype == torch.float32 and weight.dtype == torch.float32 \
                        and tuple(delta.shape) == tuple(weight.shape) == (1, sel.numel(), ht, wd, 2):
                    delta_c, weight_c = delta.contiguous(), weight.contiguous()     # (bound: alive across the launch)
                    _lib.check(L.gs_lowmem_scatter(_lib.ptr(c1), _lib.ptr(delta_c), _lib.ptr(weight_c),
                                                   _lib.ptr(net), _lib.ptr(sel), _lib.ptr(self.target), _lib.ptr(self.weight),
                                                   _lib.ptr(self.net), sel.numel(), ht, wd, st), "lowmem_scatter")
                else:
                    self.net[:, sel] = net.to(self.net.dtype)
                    self.target[:, sel] = c1 + delta.float()
                    self.weight[:, sel] = weight.float()
                self.damping[ck["uniq"]] = damping.float()
            damping = 0.2 * self.damping[idx["damping_index"]].contiguous() + EPS
            target = self.target.view(-1, self.ht, self.wd, 2).permute(0, 3, 1, 2).contiguous()
            weight = self.weight.view(-1, self.ht, self.wd, 2).permute(0, 3, 1, 2).contiguous()
            lm, ep = (1e-4, 1e-1) if ba_type == "loop" else (1e-5, 1e-2)
            # (the cached chunk index also keeps the BA's index tables, as update()'s edge index does: built by the first step
            # on this edge set -- ba_prep_kernel, 183 us at 1200 edges -- and reused by the other steps of the invocation)
            kw = {"tables": idx.setdefault("ba_tables", {})} if (BA_TABLES and getattr(self.video, "ba_accepts_tables", False)) \
                else {}
            self.video.ba(target, weight, damping, ii_all, jj_all, t0=t0, t1=t1,
                          iters=iters, lm=lm, ep=ep, motion_only=motion_only, ba_type=ba_type, **kw)
            self.video.dirty[:t] = True
