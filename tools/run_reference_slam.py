"""Run the reference's OWN objects -- `SLAM(args, cfg)` from an unmodified copy of the reference's `src/` -- on this
package's native substitutes (SURVEY 8b Boundary 4, "the acceptance test"):

  * `import droid_backends / tinycudann / lietorch / torch_scatter` inside the reference resolve to go_slam_amd
    (go_slam_amd.dropin.install()); nothing in the reference tree is edited;
  * `SLAM.__init__` builds the reference's InstantNeuS (src/InstantNeuS.py, on tcnn_compat), DroidNet, Renderer,
    DepthVideo (src/depth_video.py: shared HIP buffers), Tracker (MotionFilter + Frontend + FactorGraph + CorrBlock,
    all reference code), BundleAdjustment (Backend), MultiviewFilter, PoseTrajectoryFiller, Mapper, Mesher;
  * a synthetic RGB-D stream (a textured plane seen by a translating camera: real optical flow that is consistent
    with the depth maps and poses) is pushed through `slam.tracker(...)` for --frames frames -- the body of
    `SLAM.tracking` (slam.py:210-226) without the multi-process spin-wait --, then one `slam.ba()` (full BA,
    slam.py:63-82), one `slam.multiview_filter()` and one `slam.mapper()` (Mapper.__call__ -> optimize_map,
    src/mapping.py:60-148: reference loss + autograd through tcnn_compat), and `traj_filler` on the stream;
  * the reference's `src/InstantNeuS.py` forward + backward is compared with this package's fused pipeline on the
    same rays and parameters (--neus-check).

The reference tree is NOT part of this repository: pass --ref (default: scratch/refsrc, an untracked copy that travels
to the GPU box with the snapshot; on the build box /root/reference).  `droid.pth` is not available offline, so the
tracker runs on a randomly initialised DroidNet saved in the checkpoint format slam.load_pretrained expects: this run
proves the call contract and numerics plumbing, not tracking accuracy.

  --cpu-dryrun   build-box aid (no GPU): native ops come from the CPU oracle and the device strings in the reference
                 are rewritten to "cpu" at import time, to shake out host-side glue before spending GPU minutes.
                 Never a product path.
"""
import argparse
import importlib
import importlib.abc
import importlib.util
import json
import os
import sys
import tempfile
import time
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CPU_ONLY_LIBS = ("open3d", "cv2", "trimesh", "pyrender", "mcubes", "matplotlib", "matplotlib.pyplot", "evo", "skimage",
                 "skimage.measure", "scipy.spatial.transform", "natsort", "packaging")


def stub_cpu_libs():
    for name in CPU_ONLY_LIBS:
        if name in sys.modules:
            continue
        try:
            importlib.import_module(name)
        except Exception:
            sys.modules[name] = types.ModuleType(name)
    if "matplotlib" in sys.modules and not hasattr(sys.modules["matplotlib"], "pyplot"):
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    try:
        importlib.import_module("colorama")
    except Exception:
        col = types.ModuleType("colorama")

        class _Blank:
            def __getattr__(self, name):
                return ""
        col.Fore = col.Style = _Blank()
        sys.modules["colorama"] = col


class _CpuRewriteFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    """--cpu-dryrun only: loads `src.*` with device strings rewritten to "cpu" (the files on disk stay untouched)"""

    def __init__(self, ref):
        self.ref = ref

    def _path(self, fullname):
        parts = fullname.split(".")
        base = os.path.join(self.ref, *parts)
        if os.path.isdir(base):
            return os.path.join(base, "__init__.py"), True
        return base + ".py", False

    def find_spec(self, fullname, path=None, target=None):
        if fullname != "src" and not fullname.startswith("src."):
            return None
        p, pkg = self._path(fullname)
        if not os.path.exists(p) and not pkg:
            return None
        return importlib.util.spec_from_loader(fullname, self, origin=p, is_package=pkg)

    def create_module(self, spec):
        return None

    def exec_module(self, module):
        p, pkg = self._path(module.__name__)
        if pkg:
            module.__path__ = [os.path.dirname(p)]
        src = open(p).read() if os.path.exists(p) else ""
        for a, b in (("'cuda:0'", "'cpu'"), ('"cuda:0"', '"cpu"'), ("device='cuda'", "device='cpu'"), (".cuda()", ""),
                     ("torch.cuda.amp.autocast(enabled=True)", "torch.autocast('cpu', enabled=False)"),
                     ("torch.cuda.amp.autocast(enabled=False)", "torch.autocast('cpu', enabled=False)")):
            src = src.replace(a, b)
        module.__file__ = p
        exec(compile(src, p, "exec"), module.__dict__)


def install_cpu_oracle():
    """--cpu-dryrun: droid_backends / tinycudann from the CPU oracle (as tests/golden/gen_golden.py does)"""
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import gen_golden as G
    from oracle import droid_oracle as DO
    G.install_stubs()
    db = sys.modules["droid_backends"]
    db.ba = lambda poses, disps, intr, ds, t, w, eta, ii, jj, t0, t1, it, lm, ep, mo: DO.ba(
        poses, disps, intr, ds, t, w, eta, ii, jj, t0, t1, it, lm, ep, mo)
    db.frame_distance = lambda p, d, k, ii, jj, beta: DO.frame_distance(p, d, k, ii, jj, beta)
    import go_slam_amd.dropin as dropin
    dropin.install(droid_backends=False, tinycudann=False, lietorch=False)
    torch.cuda.empty_cache = lambda: None


def make_cfg(H_out, W_out, buffer, device, pretrained, out_dir, mode):
    """configs/go_slam.yaml of the reference, as a dict, with a small synthetic camera"""
    edge = 0
    return {
        "sync_method": "strict", "verbose": False, "dataset": "synthetic", "mode": mode, "stride": 1,
        "only_tracking": False,
        "mapping": {"device": device, "BA": False, "BA_cam_lr": 0.001, "net_lr": 0.001, "grid_lr": 0.01,
                    "w_color_loss": 2.0, "w_sdf_smooth_loss": 1.0, "w_sdf_loss": 2.0, "w_eikonal_loss": 0.1,
                    "uncertainty_weight_loss": True, "mapping_window_size": 22, "pixels": 4400, "iters": 2,
                    "post_processing_iters": 10, "decay": 0.8, "bound": [[-4.0, 4.0], [-4.0, 4.0], [-1.0, 7.0]],
                    "model": {"sdf_smooth_std": 0.005, "sdf_sparse_factor": 5, "sdf_truncation": 0.16,
                              "sdf_random_weight": 0.04, "sdf_network": {"d_in": 3, "d_out": 32},
                              "color_network": {"d_in": 3, "d_feat": 31, "d_hidden": 64, "n_layers": 2},
                              "variance_network": {"init_val": 0.2, "scale_factor": 10.0}}},
        "tracking": {"device": device, "pretrained": pretrained, "buffer": buffer, "beta": 0.75, "warmup": 8,
                     "upsample": True, "motion_filter": {"thresh": 0.0},      # random weights: keep every frame
                     "multiview_filter": {"thresh": 0.05, "visible_num": 2, "kernel_size": 1,
                                          "bound_enlarge_scale": 1.10},
                     "frontend": {"enable_loop": True, "keyframe_thresh": 0.0, "thresh": 1e4, "window": 25, "radius": 1,
                                  "nms": 1, "max_factors": 75},
                     "backend": {"thresh": 1e4, "radius": 1, "nms": 5, "loop_window": 25, "loop_thresh": 1e4,
                                 "loop_radius": 1, "loop_nms": 12}},
        "cam": {"H": H_out, "W": W_out, "fx": 0.9 * W_out, "fy": 0.9 * W_out, "cx": W_out / 2 - 0.5,
                "cy": H_out / 2 - 0.5, "png_depth_scale": 1000.0, "calibration_txt": "", "H_edge": edge, "W_edge": edge,
                "H_out": H_out, "W_out": W_out},
        "rendering": {"N_samples": 24, "N_surface": 48, "lindisp": False, "perturb": 1.0},
        "data": {"input_folder": "synthetic", "output": out_dir, "video_length": ""},
        "meshing": {"level_set": 0, "resolution": 64, "eval_rec": False, "get_largest_components": False,
                    "remove_small_geometry_threshold": 0.2, "n_points_to_eval": 200000, "mesh_threshold_to_eval": 0.05,
                    "gt_mesh_path": "", "forecast_radius": 0},
    }


class SyntheticStream:
    """A fronto-parallel textured plane at depth 3 m seen by a camera that translates along x/y and slowly towards it:
    (index, image [1,3,H,W] in [0,1], depth [H,W] in m, intrinsic [4], c2w [4,4]) -- the tuple of src/datasets.py:139."""

    def __init__(self, cfg, n, seed=7):
        c = cfg["cam"]
        self.H, self.W = c["H_out"], c["W_out"]
        self.fx, self.fy, self.cx, self.cy = c["fx"], c["fy"], c["cx"], c["cy"]
        g = torch.Generator().manual_seed(seed)
        tex = torch.rand(1, 3, 96, 128, generator=g)
        self.tex = torch.nn.functional.interpolate(tex, size=(768, 1024), mode="bicubic", align_corners=False).clamp(0, 1)
        self.n = n
        self.input_folder = "synthetic"
        self.image_timestamps = None
        poses = []
        for i in range(n):
            T = np.eye(4, dtype=np.float32)
            T[0, 3], T[1, 3], T[2, 3] = 0.06 * i, 0.02 * np.sin(0.5 * i), 0.01 * i
            poses.append(T)
        self.poses = poses
        self.Z0 = 3.0

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        T = self.poses[i]
        Z = self.Z0 - float(T[2, 3])
        v, u = torch.meshgrid(torch.arange(self.H, dtype=torch.float32), torch.arange(self.W, dtype=torch.float32),
                              indexing="ij")
        X = (u - self.cx) / self.fx * Z + float(T[0, 3])            # world point on the plane z = Z0
        Y = (v - self.cy) / self.fy * Z + float(T[1, 3])
        gx, gy = X / 4.0, Y / 3.0                                   # plane extent [-4,4] x [-3,3] -> texture [-1,1]
        img = torch.nn.functional.grid_sample(self.tex, torch.stack([gx, gy], -1)[None], mode="bilinear",
                                              padding_mode="border", align_corners=False)
        depth = torch.full((self.H, self.W), Z)
        intr = torch.tensor([self.fx, self.fy, self.cx, self.cy], dtype=torch.float32)
        return i, img, depth, intr, torch.from_numpy(T)

    def __iter__(self):
        for i in range(self.n):
            yield self[i]


def neus_file_check(slam_mod, device, log):
    """the FILE src/InstantNeuS.py (reference classes on tcnn_compat, autograd incl. autograd.grad(create_graph=True))
    vs this package's fused pipeline: same parameters, same rays -> outputs and parameter gradients"""
    import go_slam_amd.neus as neus
    from oracle import neus_oracle as NO
    RefNeuS = sys.modules["src.InstantNeuS"].InstantNeuS
    P = NO.make_params(5, grid_init=0.3, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
    model_cfg = make_cfg(64, 96, 8, device, "", "", "rgbd")["mapping"]["model"]
    ref = RefNeuS(model_cfg, bound=P["bound"].tolist(), device=device).to(device)
    mine = neus.InstantNeuS({}, P["bound"].tolist()).to(device)
    with torch.no_grad():
        for m in (ref, mine):
            m.sdf_network.encoding.encoding.params.copy_(P["grid"])
            m.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
            m.sdf_network.sdf_layer.bias.copy_(P["sdf_b"])
            m.color_network._B.copy_(P["color_B"])
            m.color_network.network.params.copy_(P["mlp"])
    g = torch.Generator().manual_seed(6)
    n = 512
    o = torch.rand(n, 3, generator=g) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    z, dist = NO.render_sample(o, d, gt, P["bound"], 24, 48, torch.rand(24, generator=g))
    o, d, z, dist = (t.to(device) for t in (o, d, z, dist))
    outs, grads = [], []
    for m in (ref, mine):
        for p in m.parameters():
            p.grad = None
        out = m(o, d, z, dist)
        loss = out["color"].mean() + out["depth"].mean() + 0.1 * out["gradient_error"].mean() + \
            out["sdf"][out["sdf"] != 100.0].mean()
        loss.backward()
        outs.append({k: v.detach().float().cpu() for k, v in out.items() if torch.is_tensor(v)})
        grads.append({k: p.grad.detach().float().cpu() for k, p in m.named_parameters() if p.grad is not None})
    rep = {}
    for k in ("color", "depth", "sdf", "normal", "weight_sum"):
        rep["out_" + k] = float((outs[0][k] - outs[1][k]).abs().max())
    for k in grads[1]:
        if k in grads[0]:
            a, b = grads[0][k], grads[1][k]
            rep["grad_rel_" + k] = float((a - b).norm() / b.norm().clamp(min=1e-12))
    log["reference_InstantNeuS_file_vs_fused_pipeline"] = rep
    return rep


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default=None)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--cpu-dryrun", action="store_true")
    ap.add_argument("--mode", default="rgbd")
    ap.add_argument("--out", default=None, help="JSON log path")
    ap.add_argument("--neus-check", action="store_true")
    ap.add_argument("--map-size", default="64x96")
    a = ap.parse_args()
    ref = a.ref or (os.path.join(ROOT, "scratch", "refsrc") if os.path.isdir(os.path.join(ROOT, "scratch", "refsrc", "src"))
                    else "/root/reference")
    assert os.path.isdir(os.path.join(ref, "src")), f"no reference tree at {ref}"
    device = "cpu" if a.cpu_dryrun else "cuda:0"
    log = {"reference_tree": ref, "device": device, "frames": a.frames, "mode": a.mode, "stages": {}}
    stub_cpu_libs()
    if a.cpu_dryrun:
        install_cpu_oracle()
        sys.meta_path.insert(0, _CpuRewriteFinder(ref))
        torch.set_num_threads(max(1, (os.cpu_count() or 8)))
    else:
        assert torch.cuda.is_available()
        import go_slam_amd.dropin as dropin
        dropin.install()
        sys.path.insert(0, ref)
    slam_mod = importlib.import_module("src.slam")
    log["stages"]["import_src_slam"] = "ok"
    H, W = (int(v) for v in a.map_size.split("x"))
    out_dir = tempfile.mkdtemp(prefix="refslam_")
    # checkpoint in the format slam.load_pretrained slices (slam.py:196-208), from the reference's own DroidNet
    torch.manual_seed(3)
    net0 = sys.modules["src.droid_net"].DroidNet()
    with torch.no_grad():                    # tame output heads: a random net must not throw the poses to infinity
        net0.update.delta[2].weight.mul_(0.02)
        net0.update.delta[2].bias.zero_()
    ckpt = os.path.join(out_dir, "droid_random.pth")
    torch.save({"module." + k: v for k, v in net0.state_dict().items()}, ckpt)
    cfg = make_cfg(H, W, max(32, a.frames + 24), device, ckpt, out_dir, a.mode)   # every frame is promoted; the trajectory filler appends batches of 16
    args = types.SimpleNamespace(device=device, make_video=False, output=out_dir)
    t0 = time.time()
    slam = slam_mod.SLAM(args, cfg)
    log["stages"]["SLAM.__init__"] = {"seconds": round(time.time() - t0, 2),
                                      "classes": {k: type(getattr(slam, k)).__module__ + "." + type(getattr(slam, k)).__name__
                                                  for k in ("mapping_net", "net", "renderer", "video", "tracker", "ba",
                                                            "multiview_filter", "traj_filler", "mapper", "mesher")}}
    if a.cpu_dryrun:                         # no fp16 autocast on the CPU: keep the feature buffers in fp32
        for name in ("fmaps", "nets", "inps"):
            setattr(slam.video, name, getattr(slam.video, name).float())
    # count the frontend's loop-closure bundle adjustments (src/frontend.py:83-87: once the window is full, every keyframe
    # runs Backend.loop_ba over ALL keyframes instead of the local update) without touching the reference's code
    loops = {"calls": 0, "last_keyframes_edges": None}
    lc = getattr(getattr(slam.tracker, "frontend", None), "loop_closing", None)
    if lc is not None:
        real_loop_ba = lc.loop_ba

        def counted_loop_ba(*args_, **kw_):
            out = real_loop_ba(*args_, **kw_)
            loops["calls"] += 1
            try:
                loops["last_keyframes_edges"] = [int(out[0]), int(out[1])]
            except Exception:
                pass
            return out
        lc.loop_ba = counted_loop_ba
    stream = SyntheticStream(cfg, a.frames)
    t0 = time.time()
    for (timestamp, image, depth, intrinsic, gt_pose) in stream:      # body of SLAM.tracking (slam.py:214-218)
        if slam.mode != "rgbd":
            depth = None
        slam.tracker(timestamp, image, depth, intrinsic, gt_pose)
    if device != "cpu":
        torch.cuda.synchronize()
    n_kf = int(slam.video.counter.value)
    poses = slam.video.poses[:n_kf].detach().cpu()
    log["stages"]["tracking"] = {"seconds": round(time.time() - t0, 2), "frames": a.frames, "keyframes": n_kf,
                                 "poses_finite": bool(torch.isfinite(poses).all()),
                                 "disps_finite": bool(torch.isfinite(slam.video.disps[:n_kf]).all()),
                                 "max_translation": float(poses[:, :3].abs().max()),
                                 "loop_closure_ba_calls": loops["calls"],
                                 "last_loop_ba_keyframes_edges": loops["last_keyframes_edges"]}
    t0 = time.time()
    slam.ba.frontend_window = min(slam.ba.frontend_window, max(1, n_kf - 2))       # (25 keyframes would be needed otherwise)
    slam.ba()
    log["stages"]["full_BA"] = {"seconds": round(time.time() - t0, 2), "last_t": int(slam.ba.last_t),
                                "poses_finite": bool(torch.isfinite(slam.video.poses[:n_kf]).all())}
    t0 = time.time()
    slam.multiview_filter()
    log["stages"]["multiview_filter"] = {"seconds": round(time.time() - t0, 2),
                                         "filtered_id": int(slam.video.filtered_id.item()),
                                         "valid_fraction": float(slam.video.mask_filtered[:max(1, n_kf)].float().mean())}
    t0 = time.time()
    before = slam.mapping_net.sdf_network.encoding.encoding.params.detach().clone()
    slam.mapper()
    after = slam.mapping_net.sdf_network.encoding.encoding.params.detach()
    log["stages"]["mapper"] = {"seconds": round(time.time() - t0, 2), "global_step": int(slam.mapper.global_step),
                               "grid_params_changed": int((before != after).sum()),
                               "params_finite": bool(torch.isfinite(after).all())}
    t0 = time.time()
    traj = slam.traj_filler(stream)
    log["stages"]["traj_filler"] = {"seconds": round(time.time() - t0, 2), "poses": list(traj.data.shape),
                                    "finite": bool(torch.isfinite(traj.data).all())}
    if a.neus_check and not a.cpu_dryrun:
        neus_file_check(slam_mod, device, log)
    try:
        import go_slam_amd._lib as L
        log["native_library"] = L.LIB_PATH
    except Exception:
        pass
    txt = json.dumps(log, indent=1)
    print(txt)
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        open(a.out, "w").write(txt)


if __name__ == "__main__":
    main()
