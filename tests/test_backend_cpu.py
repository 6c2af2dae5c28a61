"""Host-side global / loop BA driver (go_slam_amd/backend.py) against the reference's own Backend.ba
(src/backend.py:25-159), replayed by tests/golden/gen_golden.py on a fixed distance matrix -> backend_edges.npz."""
import importlib.util
import os
import types

import numpy as np
import torch

from go_slam_amd.backend import Backend, propose_backend_edges

HERE = os.path.dirname(__file__)


def _gen():
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(HERE, "golden", "gen_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    return gen


def test_backend_ba_matches_reference_golden():
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "backend_edges.npz"))
    dist = torch.from_numpy(gold["dist"])
    assert torch.equal(gen.backend_distance(), dist)
    res = gen.run_backend_cases(Backend, dist)
    assert len(res) == len(gen.BACKEND_CASES)
    for name, r in res.items():
        assert int(r["ret"]) == int(gold[f"{name}_ret"]), name
        assert np.array_equal(r["es"].numpy(), gold[f"{name}_es"]), name           # same edges, same order
        assert repr(r["calls"]) == str(gold[f"{name}_calls"]), name                # same graph calls and arguments
        assert np.array_equal(r["dirty"].numpy(), gold[f"{name}_dirty"]), name
    assert int(gold["few_ret"]) == 0 and int(gold["loop_ret"]) > 0


def test_proposal_does_not_modify_input_and_respects_cap():
    rng = np.random.default_rng(3)
    d = rng.uniform(0, 20, size=(12, 12)).astype(np.float32)
    keep = d.copy()
    es = propose_backend_edges(d, 0, 0, 12, 1, 0, 30.0, 30)
    assert np.array_equal(d, keep)
    # the cap is checked before each pick and a pick adds 2 edges: at most max_factors + 2 (reference semantics)
    assert 30 < len(es) <= 32
    assert es[:2] == [(1, 0), (0, 1)]


def test_dense_and_loop_ba_budgets(monkeypatch):
    """dense_ba / loop_ba pass the reference's factor budgets and windows to ba (src/backend.py:122-159)."""
    import go_slam_amd.backend as B
    made = []

    class FakeGraph:
        def __init__(self, video, update_op, device=None, corr_impl=None, max_factors=None, upsample=None):
            self.ii = torch.zeros(0, dtype=torch.long)
            self.kw = dict(corr_impl=corr_impl, max_factors=max_factors, upsample=upsample)
            made.append(self)
    monkeypatch.setattr(B, "FactorGraph", FakeGraph)
    video = types.SimpleNamespace(stereo=True)
    cfg = {"tracking": {"upsample": True, "beta": 0.75, "backend": {
        "thresh": 25.0, "radius": 1, "nms": 5, "loop_window": 25, "loop_thresh": 26.0, "loop_radius": 2,
        "loop_nms": 12}}}
    be = Backend(types.SimpleNamespace(update="op"), video, types.SimpleNamespace(device="cpu"), cfg)
    seen = []
    be.ba = lambda *a, **k: seen.append((a, k)) or 7
    assert be.dense_ba(3, 43, steps=5) == (40, 7)
    a, k = seen[-1]
    assert made[-1].kw == dict(corr_impl="alt", max_factors=(1 + 3 * 2) * 40, upsample=True)
    assert a[:3] == (3, 43, 5) and a[4:] == (5, 1, 25.0, 280) and k == {"motion_only": False}
    local = types.SimpleNamespace(ii=torch.arange(6), jj=torch.arange(6), age=torch.zeros(6, dtype=torch.long),
                                  net=None, target=torch.zeros(1, 6, 2, 2, 2), weight=torch.zeros(1, 6, 2, 2, 2))
    assert be.loop_ba(0, 60, steps=4, motion_only=True, local_graph=local) == (25, 7)
    a, k = seen[-1]
    assert made[-1].kw["max_factors"] == 200 and torch.equal(made[-1].ii, local.ii) and made[-1].ii is not local.ii
    assert a[4:] == (12, 2, 26.0, 200 - 6) and k == {"t_start_loop": 35, "loop": True, "motion_only": True}


def test_frontend_call_sequence_matches_reference_golden(monkeypatch):
    """Our Frontend, driven over the scripted keyframe stream of gen_golden.run_frontend_trace with the same tracing
    graph, must issue the reference Frontend's exact call sequence (src/frontend.py:48-160: age-based retirement,
    proximity proposal arguments, 4 + 2 updates, keyframe drop, loop-closure hand-off) and leave the same video
    state.  Runs with both counter flavours (plain int and multiprocessing.Value-like)."""
    import go_slam_amd.frontend as F
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "frontend_trace.npz"))
    monkeypatch.setattr(F, "FactorGraph", gen.TraceGraph)
    monkeypatch.setattr(F, "LoopClosing", gen.TraceLoop)
    for loop in (False, True):
        tag = "loop" if loop else "noloop"
        for value_counter in (True, False):
            trace, video, fe = gen.run_frontend_trace(F.Frontend, loop, value_counter)
            assert repr(trace) == str(gold[f"{tag}_trace"]), (tag, value_counter)
            assert np.allclose(video.poses.numpy(), gold[f"{tag}_poses"], rtol=0, atol=1e-6)
            assert np.allclose(video.disps.numpy(), gold[f"{tag}_disps"], rtol=1e-6, atol=0)
            assert np.array_equal(video.dirty.numpy(), gold[f"{tag}_dirty"])
            last = torch.cat([fe.last_pose, fe.last_disp.reshape(-1), fe.last_time.reshape(-1)])
            assert np.allclose(last.numpy(), gold[f"{tag}_last"], atol=1e-6)
            assert video.ready.value == int(gold[f"{tag}_ready"]) == 1
    assert "loop_ba" in str(gold["loop_trace"]) and "loop_ba" not in str(gold["noloop_trace"])
    assert "rmkf" in str(gold["loop_trace"])


def test_update_lowmem_matches_reference_golden(monkeypatch):
    """FactorGraph.update_lowmem vs the reference's (src/factor_graph.py:253-321; fixture update_lowmem.npz), kernels
    mocked on both sides: chunks of 13 source keyframes (incl. a chunk without sources), rig / stereo index arithmetic
    of the alt-corr call, per-chunk update-operator inputs, upsampling of the chunk's source keyframes, damping rows,
    BA operands and its lm / ep per ba_type, default t0 / t1, dirty flags, and the graph state left behind."""
    import go_slam_amd.factor_graph as FG
    gen = _gen()
    gold = np.load(os.path.join(HERE, "golden", "update_lowmem.npz"))
    monkeypatch.setattr(FG, "AltCorrBlock", gen.FakeAltCorr)
    for value_counter in (True, False):
        with torch.no_grad():
            log, graph, video = gen.run_lowmem(
                lambda *a, **k: FG.FactorGraph(*a, channels_last=False, **k), 1, value_counter)
        struct, nums = gen.flatten_log(log)
        assert struct == str(gold["struct"])
        assert len(nums) == int(gold["n"])
        for k, x in enumerate(nums):
            assert np.allclose(x.numpy(), gold[f"d{k}"], rtol=1e-5, atol=1e-6), k
        for name, mine in (("net", graph.net), ("target", graph.target), ("weight", graph.weight),
                           ("damping", graph.damping), ("poses", video.poses)):
            assert np.allclose(mine.float().numpy(), gold[name], rtol=1e-5, atol=1e-6), name
        assert np.array_equal(video.dirty.numpy(), gold["dirty"])
