"""Local-window tracking driver (mirrors src/frontend.py:9-160): decides, per new keyframe, which edges enter the
factor graph and how many update-operator + BA iterations run -- the caller of the whole T hot path
(FactorGraph.update = reproject -> corr lookup -> update operator -> dense BA).

The sequence of graph calls and their arguments is the reference's (pinned by a golden call trace); what is left on
the host is only what must be: one scalar read per keyframe (the keyframe-distance test steers control flow) and the
edge bookkeeping, which FactorGraph keeps as cached host copies.  The work of one keyframe is split into named
stages below instead of the reference's two monolithic methods.
"""
import contextlib
from dataclasses import dataclass

import torch

from .backend import Backend as LoopClosing
from .factor_graph import FactorGraph


def keyframe_count(video):
    """video.counter is an int in the host mirror and a multiprocessing.Value in the reference (depth_video.py:27)."""
    c = video.counter
    return int(getattr(c, "value", c))


def set_keyframe_count(video, n):
    if hasattr(video.counter, "value"):
        video.counter.value = n
    else:
        video.counter = n


def _lock(video):
    get = getattr(video, "get_lock", None)
    return get() if get is not None else contextlib.nullcontext()


@dataclass
class FrontendOptions:
    """cfg['tracking'] entries the frontend reads (configs/go_slam.yaml: tracking.frontend.*)"""
    warmup: int
    upsample: bool
    beta: float
    max_factors: int
    nms: int
    keyframe_thresh: float
    window: int
    thresh: float
    radius: int
    enable_loop: bool
    max_age: int = 25          # updates after which an edge is retired to the inactive set
    iters_first: int = 4       # update iterations before the keyframe decision
    iters_second: int = 2      # ... after it (also the loop-closure step count)

    @classmethod
    def from_cfg(cls, cfg):
        trk, fe = cfg["tracking"], cfg["tracking"]["frontend"]
        return cls(warmup=trk["warmup"], upsample=trk["upsample"], beta=trk["beta"], max_factors=fe["max_factors"],
                   nms=fe["nms"], keyframe_thresh=fe["keyframe_thresh"], window=fe["window"], thresh=fe["thresh"],
                   radius=fe["radius"], enable_loop=fe["enable_loop"])


class Frontend:
    def __init__(self, net, video, args, cfg):
        self.video = video
        self.update_op = net.update
        self.opt = o = FrontendOptions.from_cfg(cfg)
        self.verbose = cfg.get("verbose", False)
        # the reference's attribute names, for callers that read them
        self.warmup, self.upsample, self.beta = o.warmup, o.upsample, o.beta
        self.frontend_max_factors, self.frontend_nms, self.frontend_window = o.max_factors, o.nms, o.window
        self.keyframe_thresh, self.frontend_thresh, self.frontend_radius = o.keyframe_thresh, o.thresh, o.radius
        self.enable_loop, self.max_age, self.iters1, self.iters2 = o.enable_loop, o.max_age, o.iters_first, o.iters_second
        self.loop_closing = LoopClosing(net, video, args, cfg)
        self.last_loop_t = -1
        self.device = args.device
        self.graph = FactorGraph(video, net.update, device=args.device, corr_impl="volume", max_factors=o.max_factors,
                                 upsample=o.upsample)
        self.t0 = self.t1 = 0            # local optimisation window [t0, t1)
        self.is_initialized = False
        self.count = 0

    # ---- stages of one keyframe (src/frontend.py:48-104) ---------------------------------------------------------
    def _refine(self, n, **window):
        for _ in range(n):
            self.graph.update(use_inactive=True, **window)

    def _retire_and_propose(self):
        """old edges go to the inactive set; proximity edges between [t1-5, counter) and [t1-window, counter)"""
        g, o = self.graph, self.opt
        if g.corr is not None:
            g.rm_factors(g.age > o.max_age, store=True)
        g.add_proximity_factors(self.t1 - 5, max(self.t1 - o.window, 0), rad=o.radius, nms=o.nms, thresh=o.thresh,
                                beta=o.beta, remove=True)

    def _seed_depth_from_sensor(self, k):
        v = self.video
        v.disps[k] = torch.where(v.disps_sens[k] > 0, v.disps_sens[k], v.disps[k])

    def _moved_enough(self):
        """the one host read of a keyframe: mean flow between the two newest keyframes vs keyframe_thresh"""
        ar = getattr(self, "_arange", None)     # (views of a device arange: two index lists would be two blocking uploads)
        if ar is None or ar.numel() <= self.t1:
            ar = self._arange = torch.arange(2 * (self.t1 + 64), device=torch.device(self.device))
        d = self.video.distance(ar[self.t1 - 3:self.t1 - 2], ar[self.t1 - 2:self.t1 - 1], beta=self.opt.beta,
                                bidirectional=True)
        return float(d) >= self.opt.keyframe_thresh

    def _drop_previous_keyframe(self):
        self.graph.rm_keyframe(self.t1 - 2)
        with _lock(self.video):
            set_keyframe_count(self.video, keyframe_count(self.video) - 1)
            self.t1 -= 1

    def _close_loop_or_refine(self):
        cur_t = keyframe_count(self.video)
        if self.opt.enable_loop and cur_t > self.opt.window:
            n_kf, n_edge = self.loop_closing.loop_ba(t_start=0, t_end=cur_t, steps=self.opt.iters_second,
                                                     motion_only=False, local_graph=self.graph)
            if self.verbose:
                print(f"Loop BA: [0, {cur_t}]; {n_kf} KFs, {n_edge} edges, last loop KF {self.last_loop_t}")
            self.last_loop_t = cur_t
        else:
            self._refine(self.opt.iters_second, t0=None, t1=None)

    def _seed_next_frame(self):
        v = self.video
        v.poses[self.t1] = v.poses[self.t1 - 1]
        v.disps[self.t1] = v.disps[self.t1 - 1].mean()
        c = getattr(self.graph, "_eidx", None)          # (the cached edge index knows the oldest source keyframe: no device read)
        lo = c["ii_min"] if c is not None and c["tens"][0] is self.graph.ii and "ii_min" in c else int(self.graph.ii.min())
        v.dirty[lo:self.t1] = True

    @torch.no_grad()
    def _update(self):
        self.count += 1
        self.t1 += 1
        self._retire_and_propose()
        self._seed_depth_from_sensor(self.t1 - 1)
        self._refine(self.opt.iters_first, t0=None, t1=None)
        if self._moved_enough():
            self._close_loop_or_refine()
        else:                                   # too little motion: the previous keyframe is dropped
            self._drop_previous_keyframe()
        self._seed_next_frame()

    # ---- bootstrap on the first `warmup` keyframes (src/frontend.py:106-142) --------------------------------------
    @torch.no_grad()
    def _initialize(self):
        v, g = self.video, self.graph
        self.t0, self.t1 = 0, keyframe_count(v)
        g.add_neighborhood_factors(self.t0, self.t1, r=3)
        self._refine(8, t0=1, t1=None)
        g.add_proximity_factors(t0=0, t1=0, rad=2, nms=2, thresh=self.opt.thresh, remove=False)
        self._refine(8, t0=1, t1=None)
        last = self.t1 - 1
        v.poses[self.t1] = v.poses[last].clone()
        v.disps[self.t1] = v.disps[self.t1 - 4:self.t1].mean()
        self.is_initialized = True
        self.last_pose, self.last_disp, self.last_time = v.poses[last].clone(), v.disps[last].clone(), v.timestamp[last].clone()
        with _lock(v):
            if hasattr(v, "ready"):
                v.ready.value = 1
            v.dirty[:self.t1] = True
        g.rm_factors(g.ii < self.opt.warmup - 4, store=True)

    def __call__(self):
        """main update (src/frontend.py:144-160): initialise at the warm-up count, then one _update per new keyframe"""
        n = keyframe_count(self.video)
        if not self.is_initialized:
            if n == self.opt.warmup:
                self._initialize()
        elif self.t1 < n:
            self._update()
