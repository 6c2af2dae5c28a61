"""Local-window tracking driver (mirrors src/frontend.py:9-160): decides, per new keyframe, which edges enter the
factor graph and how many update-operator + BA iterations run -- the caller of the whole T hot path
(FactorGraph.update = reproject -> corr lookup -> update operator -> dense BA).

The sequence of graph calls and their arguments is the reference's; what is left on the host is only what must be:
one scalar read per keyframe (the keyframe-distance test steers control flow) and the edge bookkeeping, which
FactorGraph keeps as cached host copies.
"""
import contextlib

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


class Frontend:
    def __init__(self, net, video, args, cfg):
        self.video = video
        self.update_op = net.update
        trk = cfg["tracking"]
        fe = trk["frontend"]
        self.warmup = trk["warmup"]
        self.upsample = trk["upsample"]
        self.beta = trk["beta"]
        self.verbose = cfg.get("verbose", False)
        self.frontend_max_factors = fe["max_factors"]
        self.frontend_nms = fe["nms"]
        self.keyframe_thresh = fe["keyframe_thresh"]
        self.frontend_window = fe["window"]
        self.frontend_thresh = fe["thresh"]
        self.frontend_radius = fe["radius"]
        self.enable_loop = fe["enable_loop"]
        self.loop_closing = LoopClosing(net, video, args, cfg)
        self.last_loop_t = -1
        self.graph = FactorGraph(video, net.update, device=args.device, corr_impl="volume",
                                 max_factors=self.frontend_max_factors, upsample=self.upsample)
        self.t0 = 0                      # local optimisation window [t0, t1)
        self.t1 = 0
        self.is_initialized = False
        self.count = 0
        self.max_age = 25
        self.iters1 = 4
        self.iters2 = 2

    @torch.no_grad()
    def _update(self):
        """add edges for the newest keyframe, run the update operator, keep or drop the keyframe
        (src/frontend.py:48-104)."""
        v, g = self.video, self.graph
        self.count += 1
        self.t1 += 1
        if g.corr is not None:
            g.rm_factors(g.age > self.max_age, store=True)
        # edges between [t1-5, counter) and [t1-window, counter)
        g.add_proximity_factors(self.t1 - 5, max(self.t1 - self.frontend_window, 0), rad=self.frontend_radius,
                                nms=self.frontend_nms, thresh=self.frontend_thresh, beta=self.beta, remove=True)
        k = self.t1 - 1
        v.disps[k] = torch.where(v.disps_sens[k] > 0, v.disps_sens[k], v.disps[k])
        for _ in range(self.iters1):
            g.update(t0=None, t1=None, use_inactive=True)
        # too little motion between the last two keyframes -> drop the older one
        d = v.distance([self.t1 - 3], [self.t1 - 2], beta=self.beta, bidirectional=True)
        if float(d) < self.keyframe_thresh:
            g.rm_keyframe(self.t1 - 2)
            with _lock(v):
                set_keyframe_count(v, keyframe_count(v) - 1)
                self.t1 -= 1
        else:
            cur_t = keyframe_count(v)
            if self.enable_loop and cur_t > self.frontend_window:
                n_kf, n_edge = self.loop_closing.loop_ba(t_start=0, t_end=cur_t, steps=self.iters2,
                                                         motion_only=False, local_graph=g)
                if self.verbose:
                    print(f"Loop BA: [0, {cur_t}]; {n_kf} KFs, {n_edge} edges, last loop KF {self.last_loop_t}")
                self.last_loop_t = cur_t
            else:
                for _ in range(self.iters2):
                    g.update(t0=None, t1=None, use_inactive=True)
        # initial guess for the next frame
        v.poses[self.t1] = v.poses[self.t1 - 1]
        v.disps[self.t1] = v.disps[self.t1 - 1].mean()
        v.dirty[int(g.ii.min()):self.t1] = True

    @torch.no_grad()
    def _initialize(self):
        """bootstrap on the first `warmup` keyframes (src/frontend.py:106-142)."""
        v, g = self.video, self.graph
        self.t0 = 0
        self.t1 = keyframe_count(v)
        g.add_neighborhood_factors(self.t0, self.t1, r=3)
        for _ in range(8):
            g.update(t0=1, t1=None, use_inactive=True)
        g.add_proximity_factors(t0=0, t1=0, rad=2, nms=2, thresh=self.frontend_thresh, remove=False)
        for _ in range(8):
            g.update(t0=1, t1=None, use_inactive=True)
        v.poses[self.t1] = v.poses[self.t1 - 1].clone()
        v.disps[self.t1] = v.disps[self.t1 - 4:self.t1].mean()
        self.is_initialized = True
        self.last_pose = v.poses[self.t1 - 1].clone()
        self.last_disp = v.disps[self.t1 - 1].clone()
        self.last_time = v.timestamp[self.t1 - 1].clone()
        with _lock(v):
            if hasattr(v, "ready"):
                v.ready.value = 1
            v.dirty[:self.t1] = True
        g.rm_factors(g.ii < self.warmup - 4, store=True)

    def __call__(self):
        """main update (src/frontend.py:144-160)"""
        n = keyframe_count(self.video)
        if not self.is_initialized and n == self.warmup:
            self._initialize()
        elif self.is_initialized and self.t1 < n:
            self._update()
