"""Global / loop-closure bundle adjustment driver (mirrors src/backend.py:7-159): proposes the edge set of a
[t_start, t_end) window from the frame-distance matrix, builds an alt-correlation FactorGraph over it and runs
`update_lowmem`.

What differs from the reference is where the proposal runs.  The reference keeps the distance matrix on the GPU and
walks it from Python, reading one element per candidate with `.item()` and writing the NMS windows back with tiny
slice kernels (backend.py:62-94): thousands of host syncs for a 200-keyframe window.  Here the matrix (t^2 floats,
160 kB at t = 200) crosses PCIe ONCE after the two frame_distance launches and the greedy NMS is a NumPy loop on the
host; the result is the same edge list in the same order.
"""
import numpy as np
import torch

from .factor_graph import FactorGraph


def propose_backend_edges(d, t_start, t_start_loop, t_end, radius, nms, thresh, max_factors, stereo=False, loop=False):
    """Edge proposal of Backend.ba (src/backend.py:34-94) on a host copy of the distance matrix.

    d: float32 numpy [t_end - t_start_loop, t_end - t_start], raw bidirectional frame distances (not modified).
    Returns the (i, j) list in the reference's order: local window first, then candidates by increasing distance
    with (2 nms + 1)^2 non-maximum suppression; in loop mode a candidate only counts if more than half of its 3x3
    neighbourhood is below `thresh` in the RAW matrix, and then contributes that neighbourhood (one direction)."""
    ilen, jlen = t_end - t_start_loop, t_end - t_start
    rawd = np.array(d, dtype=np.float32, copy=True).reshape(ilen, jlen)
    d = rawd.copy()
    ix = np.arange(t_start_loop, t_end)[:, None]
    jx = np.arange(t_start, t_end)[None, :]
    d[ix - radius < jx] = np.inf
    d[d > thresh] = np.inf

    def suppress(di, dj):
        d[max(0, di - nms):min(ilen, di + nms + 1), max(0, dj - nms):min(jlen, dj + nms + 1)] = np.inf

    es = []
    for i in range(t_start_loop, t_end):                         # local window [i - radius, i)
        if stereo and not loop:
            es.append((i, i))
            d[i - t_start_loop, i - t_start] = np.inf
        for j in range(max(i - radius, t_start_loop), i):
            es += [(i, j), (j, i)]
            d[i - t_start_loop, j - t_start] = np.inf
            suppress(i - t_start_loop, j - t_start)
    flat = d.reshape(-1)
    order = np.argsort(flat, kind="stable")                      # distance from small to big
    order = order[flat[order] <= thresh]
    nb = 1
    for k in order.tolist():
        di, dj = k // jlen, k % jlen
        if d[di, dj] > thresh:                                   # suppressed by an earlier pick
            continue
        if len(es) > max_factors:
            break
        i, j = t_start_loop + di, t_start + dj
        if loop:
            sub, hits = [], 0
            for si in range(max(i - nb, t_start_loop), min(i + nb + 1, t_end)):
                for sj in range(max(j - nb, t_start), min(j + nb + 1, t_end)):
                    if rawd[si - t_start_loop, sj - t_start] <= thresh:
                        hits += 1
                        if si != sj:
                            sub.append((si, sj))
            if hits > int(((nb * 2 + 1) ** 2) * 0.5):
                es += sub
        else:
            es += [(i, j), (j, i)]                               # bidirectional
        suppress(di, dj)
    return es


class Backend:
    def __init__(self, net, video, args, cfg):
        self.video = video
        self.device = args.device
        self.update_op = net.update
        trk = cfg["tracking"]
        self.upsample = trk["upsample"]
        self.beta = trk["beta"]
        be = trk["backend"]
        self.backend_thresh, self.backend_radius, self.backend_nms = be["thresh"], be["radius"], be["nms"]
        self.backend_loop_window, self.backend_loop_thresh = be["loop_window"], be["loop_thresh"]
        self.backend_loop_radius, self.backend_loop_nms = be["loop_radius"], be["loop_nms"]

    def _graph(self, max_factors):
        return FactorGraph(self.video, self.update_op, device=self.device, corr_impl="alt", max_factors=max_factors,
                           upsample=self.upsample)

    @torch.no_grad()
    def ba(self, t_start, t_end, steps, graph, nms, radius, thresh, max_factors, t_start_loop=None, loop=False,
           motion_only=False):
        """main update (src/backend.py:25-120); returns the number of edges optimised (0 if fewer than 3)."""
        if t_start_loop is None or not loop:
            t_start_loop = t_start
        assert t_start_loop >= t_start, f"short: {t_start_loop}, long: {t_start}."
        vdev = torch.device(self.device)         # (built on the device: a host grid is two blocking uploads per call)
        ii, jj = torch.meshgrid(torch.arange(t_start_loop, t_end, device=vdev), torch.arange(t_start, t_end, device=vdev),
                                indexing="ij")
        d = self.video.distance(ii.reshape(-1), jj.reshape(-1), beta=self.beta)
        stereo = bool(getattr(self.video, "stereo", False))
        if d.is_cuda and (t_end - t_start_loop) * (t_end - t_start) <= 512 * 512:
            # proposal on the GPU (csrc/edge_nms.hip): the matrix stays in HBM, one int is read back
            e = FactorGraph.propose_edges_on_device(d, None, t_start_loop, t_start, t_end, radius, nms, thresh, thresh,
                                                    max_factors, stereo and not loop, t_start_loop, loop)
            if e.shape[0] < 3:
                return 0
        else:
            d = d.detach().float().cpu().numpy().reshape(t_end - t_start_loop, t_end - t_start)   # one D2H copy
            es = propose_backend_edges(d, t_start, t_start_loop, t_end, radius, nms, thresh, max_factors, stereo=stereo,
                                       loop=loop)
            if len(es) < 3:
                return 0
            e = torch.tensor(es, dtype=torch.long, device=self.device)
        graph.add_factors(e[:, 0], e[:, 1], remove=True)
        edge_num = len(graph.ii)
        # the start pose is fixed to avoid drift: t_start_loop, not t_start (src/backend.py:100-109)
        graph.update_lowmem(t0=t_start_loop + 1, t1=t_end, iters=2, use_inactive=False, steps=steps, max_t=t_end,
                            ba_type="dense", motion_only=motion_only)
        graph.clear_edges()
        self.video.dirty[t_start:t_end] = True
        return edge_num

    @torch.no_grad()
    def dense_ba(self, t_start, t_end, steps=6, motion_only=False):
        """full BA over [t_start, t_end) (src/backend.py:122-136)."""
        radius = self.backend_radius
        n = t_end - t_start
        max_factors = (int(bool(getattr(self.video, "stereo", False))) + (radius + 2) * 2) * n
        graph = self._graph(max_factors)
        n_edges = self.ba(t_start, t_end, steps, graph, self.backend_nms, radius, self.backend_thresh, max_factors,
                          motion_only=motion_only)
        return n, n_edges

    @torch.no_grad()
    def loop_ba(self, t_start, t_end, steps=6, motion_only=False, local_graph=None):
        """loop-closure BA over the last `loop_window` keyframes against all of [t_start, t_end)
        (src/backend.py:139-159); `local_graph`'s edges and state are carried over."""
        window = self.backend_loop_window
        max_factors = 8 * window
        t_start_loop = max(0, t_end - window)
        graph = self._graph(max_factors)
        if local_graph is not None:
            for key in ("ii", "jj", "age", "net", "target", "weight"):
                val = getattr(local_graph, key)
                if val is not None:
                    setattr(graph, key, val.clone())
        left = max_factors - len(graph.ii)
        n_edges = self.ba(t_start, t_end, steps, graph, self.backend_loop_nms, self.backend_loop_radius,
                          self.backend_loop_thresh, left, t_start_loop=t_start_loop, loop=True,
                          motion_only=motion_only)
        return t_end - t_start_loop, n_edges
