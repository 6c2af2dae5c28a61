"""SURVEY 8b Boundary 4 -- the `src/slam.py` import contract, on the build box (needs /root/reference; skipped elsewhere,
e.g. on the GPU box where the reference tree does not exist).

1. With `go_slam_amd.dropin.install()` the UNMODIFIED reference tree imports: `src.slam` and every sibling it pulls in
   resolve `droid_backends`, `tinycudann`, `lietorch` and `torch_scatter` to this package (the absent pure-CPU
   libraries -- open3d, cv2, trimesh, pyrender, mcubes, matplotlib, colorama, evo: mesh extraction, viewers, logging
   colours, all out of scope -- are stubbed as empty modules for the import).
2. The reference's own host classes construct on top of the substitutes on the CPU (no kernel runs at construction):
   `DepthVideo(cfg, args)`, `FactorGraph(video, update_op, ...)`, `DroidNet()` with the checkpoint's sub-module names,
   `PoseTrajectoryFiller`; the reference's reprojection runs on the lietorch shim.
3. This package's mirrors take the constructor signatures slam.py uses (same parameter names as the reference's
   classes), so swapping the sibling imports needs no call-site edits.
"""
import importlib
import inspect
import os
import sys
import types

import pytest
import torch

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "src")), reason="reference tree not present")

CPU_ONLY_LIBS = ("open3d", "cv2", "trimesh", "pyrender", "mcubes", "matplotlib", "matplotlib.pyplot", "evo")


@pytest.fixture(scope="module")
def ref():
    """import the reference package `src` with the native modules substituted"""
    import go_slam_amd.dropin as dropin
    saved = dict(sys.modules)
    dropin.install()
    for name in CPU_ONLY_LIBS:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                sys.modules[name] = types.ModuleType(name)
    if "matplotlib" in sys.modules and not hasattr(sys.modules["matplotlib"], "pyplot"):
        sys.modules["matplotlib"].pyplot = sys.modules["matplotlib.pyplot"]
    if "colorama" not in sys.modules:
        try:
            importlib.import_module("colorama")
        except Exception:
            col = types.ModuleType("colorama")
            col.Fore = types.SimpleNamespace(**{c: "" for c in ("RED", "GREEN", "YELLOW", "BLUE", "MAGENTA", "CYAN", "WHITE")})
            col.Style = types.SimpleNamespace(RESET_ALL="", BRIGHT="", DIM="", NORMAL="")
            sys.modules["colorama"] = col
    sys.path.insert(0, REF)
    try:
        for k in [k for k in sys.modules if k == "src" or k.startswith("src.")]:
            del sys.modules[k]
        slam = importlib.import_module("src.slam")
        yield slam
    finally:
        sys.path.remove(REF)
        for k in [k for k in sys.modules if k not in saved]:
            del sys.modules[k]


def test_unmodified_reference_slam_imports_on_the_substitutes(ref):
    import go_slam_amd.droid_backends as db
    from go_slam_amd import lietorch_shim
    from go_slam_amd.neus import tcnn_compat
    assert sys.modules["droid_backends"] is db and sys.modules["tinycudann"] is tcnn_compat
    assert sys.modules["lietorch"] is lietorch_shim
    # the reference's own modules picked the substitutes up
    assert sys.modules["src.depth_video"].droid_backends is db
    assert sys.modules["src.modules.corr"].droid_backends is db
    assert sys.modules["src.InstantNeuS"].tcnn is tcnn_compat
    assert sys.modules["src.slam"].SE3 is lietorch_shim.SE3
    for name in ("ba", "frame_distance", "projmap", "depth_filter", "iproj", "corr_index_forward",
                 "corr_index_backward", "altcorr_forward", "altcorr_backward"):         # src/lib/droid.cpp:237-250
        assert callable(getattr(db, name)), name
    for cls in ("SLAM", "Tracker", "BundleAdjustment"):
        assert hasattr(ref, cls)


def _cfg():
    return {"mode": "rgbd", "cam": {"H_out": 64, "W_out": 96}, "tracking": {"buffer": 12}}


def test_reference_host_classes_construct_on_the_substitutes(ref):
    """the reference's DepthVideo / FactorGraph / DroidNet / trajectory filler built on the drop-in modules (CPU)"""
    args = types.SimpleNamespace(device="cpu")
    RV = sys.modules["src.depth_video"].DepthVideo
    video = RV(_cfg(), args)
    assert video.counter.value == 0 and tuple(video.disps.shape) == (12, 8, 12)
    net = sys.modules["src.droid_net"].DroidNet()
    names = {n.split(".")[0] + "." + n.split(".")[1] for n, _ in net.named_parameters() if n.startswith("update.")}
    assert {"update.corr_encoder", "update.flow_encoder", "update.weight", "update.delta", "update.gru",
            "update.agg"} <= names                                                      # slam.py:196-208 slices these
    RG = sys.modules["src.factor_graph"].FactorGraph
    graph = RG(video, net.update, device="cpu", corr_impl="alt", max_factors=48, upsample=True)
    assert graph.ht == 8 and graph.wd == 12
    filler = sys.modules["src.trajectory_filler"].PoseTrajectoryFiller(net=net, video=video, device="cpu")
    assert filler.video is video
    # (the reference's MotionFilter / InstantNeuS constructors call .cuda() / torch.cuda.device: GPU only)
    # reference reprojection on the lietorch shim: identity poses map the grid onto itself
    ii = torch.tensor([0, 1]); jj = torch.tensor([1, 0])
    video.intrinsics[:] = torch.tensor([50.0, 50.0, 6.0, 4.0])
    pops = sys.modules["src.geom.projective_ops"]
    Gs = sys.modules["lietorch"].SE3(video.poses[None])
    coords, valid = pops.projective_transform(Gs, video.disps[None], video.intrinsics[None], ii, jj)
    ys, xs = torch.meshgrid(torch.arange(8.0), torch.arange(12.0), indexing="ij")      # (pops.coords_grid defaults to cuda)
    torch.testing.assert_close(coords[0, 0], torch.stack([xs, ys], -1), rtol=0, atol=1e-4)


MIRRORS = {  # reference module -> (class, this package's module)
    "src.depth_video": ("DepthVideo", "go_slam_amd.depth_video"),
    "src.factor_graph": ("FactorGraph", "go_slam_amd.factor_graph"),
    "src.frontend": ("Frontend", "go_slam_amd.frontend"),
    "src.backend": ("Backend", "go_slam_amd.backend"),
    "src.motion_filter": ("MotionFilter", "go_slam_amd.motion_filter"),
    "src.multiview_filter": ("MultiviewFilter", "go_slam_amd.multiview_filter"),
    "src.trajectory_filler": ("PoseTrajectoryFiller", "go_slam_amd.trajectory_filler"),
    "src.droid_net": ("DroidNet", "go_slam_amd.droid_net"),
    "src.InstantNeuS": ("InstantNeuS", "go_slam_amd.neus.instant_neus"),
}


@pytest.mark.parametrize("ref_mod", sorted(MIRRORS))
def test_mirror_constructors_accept_the_reference_call(ref, ref_mod):
    """every positional / keyword argument the reference class takes is accepted, in the same position, by the mirror"""
    cls, mine = MIRRORS[ref_mod]
    r = getattr(sys.modules[ref_mod], cls)
    m = getattr(importlib.import_module(mine), cls)
    rp = [p for p in inspect.signature(r.__init__).parameters.values() if p.name != "self"]
    mp = [p for p in inspect.signature(m.__init__).parameters.values() if p.name != "self"]
    if cls == "DepthVideo":                       # one polymorphic constructor: (cfg, args) or (h8, w8, ...)
        assert len(rp) == 2 and len(mp) >= 2
        v = m(_cfg(), types.SimpleNamespace(device="cpu"))
        assert v.ht == 64 and v.wd == 96 and tuple(v.disps.shape) == (12, 8, 12) and hasattr(v.counter, "value")
        return
    has_var_kw = any(p.kind is inspect.Parameter.VAR_KEYWORD for p in mp)
    mnames = [p.name for p in mp]
    for i, p in enumerate(rp):
        if p.kind in (inspect.Parameter.VAR_POSITIONAL, inspect.Parameter.VAR_KEYWORD):
            continue
        assert p.name in mnames or has_var_kw, f"{cls}: the mirror lacks parameter `{p.name}`"
        if p.default is inspect.Parameter.empty and p.name in mnames:
            assert mnames.index(p.name) == i, f"{cls}: positional parameter `{p.name}` is at a different position"
