"""Host-side index logic of FactorGraph.update (CPU): the cached edge index must equal what the
reference derives per update (src/factor_graph.py:213-240) and must refresh when an edge list changes."""
import types

import torch

from go_slam_amd.factor_graph import FactorGraph


def _graph():
    video = types.SimpleNamespace(ht=6, wd=8, disps=torch.ones(16, 6, 8))
    g = FactorGraph(video, update_op=None, device="cpu")
    g.ii = torch.tensor([3, 3, 4, 5, 5, 5, 6])
    g.jj = torch.tensor([4, 5, 3, 3, 4, 6, 5])
    g.ii_inac = torch.tensor([0, 1, 2, 2, 3])
    g.jj_inac = torch.tensor([1, 2, 1, 3, 2])
    return g


def _reference(g, t0, t1, use_inactive):
    if t0 is None:
        t0 = max(1, int(g.ii.min()) + 1)
    t0 = max(1, t0)
    if t1 is None:
        t1 = max(int(g.ii.max()), int(g.jj.max())) + 1
    if use_inactive:
        m = (g.ii_inac >= t0 - 3) & (g.jj_inac >= t0 - 3)
        ii = torch.cat([g.ii_inac[m], g.ii])
        jj = torch.cat([g.jj_inac[m], g.jj])
    else:
        m, ii, jj = None, g.ii, g.jj
    dix = torch.unique(torch.cat([torch.arange(t0, t1), ii]), sorted=True)
    return t0, t1, m, ii, jj, dix


def test_edge_index_matches_reference_logic():
    g = _graph()
    for t0, t1, inac in [(None, None, True), (None, None, False), (2, 9, True), (0, None, True)]:
        c = g._edge_index(t0, t1, inac)
        r0, r1, m, ii, jj, dix = _reference(g, t0, t1, inac)
        assert (c["t0"], c["t1"]) == (r0, r1)
        assert torch.equal(c["ii"], ii) and torch.equal(c["jj"], jj)
        assert torch.equal(c["damping_index"], dix)
        if inac:
            assert torch.equal(c["sel"], torch.nonzero(m).reshape(-1))
        uniq, ix = torch.unique(g.ii, sorted=True, return_inverse=True)
        # damping rows (gs_damping_rows): inv[k] = the operator's eta row for frame damping_index[k], -1 if it has none
        inv = c["damping_inv"]
        assert inv.dtype == torch.int32 and inv.numel() == dix.numel()
        for k, f in enumerate(dix.tolist()):
            hit = torch.nonzero(uniq == f).reshape(-1)
            assert int(inv[k]) == (int(hit[0]) if hit.numel() else -1)
        assert set(uniq.tolist()) <= set(dix.tolist())         # every eta row lands in the buffer through some k
        seg = c["seg"]
        assert torch.equal(seg["uniq"], uniq) and torch.equal(seg["ix"], ix)
        for s in range(seg["n"]):
            edges = seg["order"][seg["offsets"][s]:seg["offsets"][s + 1]].long()
            assert torch.equal(torch.sort(edges).values, torch.nonzero(ix == s).reshape(-1))


def test_edge_index_cache_invalidation():
    g = _graph()
    a = g._edge_index(None, None, True)
    assert g._edge_index(None, None, True) is a                 # steady state: cached
    assert g._edge_index(None, None, False) is not a            # different arguments
    b = g._edge_index(None, None, True)
    g.ii[g.ii >= 5] -= 1                                        # in-place edit (rm_keyframe style)
    c = g._edge_index(None, None, True)
    assert c is not b and torch.equal(c["seg"]["uniq"], torch.unique(g.ii))
    g.jj = torch.cat([g.jj[:-1], torch.tensor([9])])            # replaced tensor (add/rm_factors style)
    d = g._edge_index(None, None, True)
    assert d is not c and d["t1"] == 10
    g.ii_inac = g.ii_inac[:-1]
    g.jj_inac = g.jj_inac[:-1]
    e = g._edge_index(None, None, True)
    assert e is not d and e["ii"].numel() == d["ii"].numel() - 1


def test_masked_sdf_error_equals_gather_formulation():
    """InstantNeuS.compute_sdf_error masks rays without depth instead of gathering the valid ones
    (reference src/InstantNeuS.py:372-400 uses boolean indexing); both must give the same two numbers."""
    from go_slam_amd.neus import InstantNeuS
    from oracle import neus_oracle as NO
    g = torch.Generator().manual_seed(5)
    n, s = 37, 72
    gt = torch.rand(n, generator=g) * 3 + 0.5
    gt[torch.rand(n, generator=g) < 0.3] = 0.0
    z = torch.sort(torch.rand(n, s, generator=g) * 4, dim=1).values
    sdf = torch.randn(n, s, generator=g) * 0.2
    model = InstantNeuS({}, [[-1.0, 1.0]] * 3, device="cpu")
    e, f = model.compute_sdf_error(sdf, z, gt)
    re, rf = NO.compute_sdf_error(sdf[gt > 0], z[gt > 0], gt[gt > 0].reshape(-1, 1), model.sdf_truncation,
                                  model.sdf_sparse_factor)
    torch.testing.assert_close(e, re, rtol=1e-5, atol=1e-7)
    torch.testing.assert_close(f, rf, rtol=1e-5, atol=1e-7)


def test_mfma_fragment_packers_match_their_documented_layouts():
    """Host-side packing of weights into MFMA A-fragments (gs_conv1x1, gs_conv3x3_head, gs_mlp_backward):
    element (frag, lane l, e) must be M[32 mt + (l & 31)][16 ks + 8 (l >> 5) + e] of the documented matrix."""
    from go_slam_amd.droid_net import pack_1x1_weight, pack_head_weight
    from go_slam_amd.neus.instant_neus import _pack_mlp_fragments
    g = torch.Generator().manual_seed(9)
    # 1x1: W [N,K,1,1], K = 196 -> 13 k-steps, zero padded
    W = torch.randn(64, 196, 1, 1, generator=g)
    P = pack_1x1_weight(W).float()                         # [N/32, KS, 64, 8]
    Wp = torch.zeros(64, 208)
    Wp[:, :196] = W.reshape(64, 196).half().float()
    for nb, ks, l, e in [(0, 0, 0, 0), (1, 12, 63, 7), (1, 5, 37, 3), (0, 12, 31, 4)]:
        assert P[nb, ks, l, e] == Wp[32 * nb + (l & 31), 16 * ks + 8 * (l >> 5) + e]
    # 3x3 heads: W [O,128,3,3] -> wpack[ks][l][e] = W[o][16 ks + 8 (l>>5) + e][ky][kx], (l & 31) = (3 ky + kx) O + o
    for O in (1, 2):
        W = torch.randn(O, 128, 3, 3, generator=g)
        P = pack_head_weight(W).float()
        Wh = W.half().float()
        for ks, l, e in [(0, 0, 0), (7, 40, 7), (3, 17, 2), (5, 9 * O - 1, 1), (2, 31, 0), (6, 63, 5)]:
            col = l & 31
            want = Wh[col % O, 16 * ks + 8 * (l >> 5) + e, (col // O) // 3, (col // O) % 3] if col < 9 * O else 0.0
            assert P[ks, l, e] == want, (O, ks, l, e)
    # colour-MLP backward: 40 fragments of W1 | W2 | W3^T | W2^T | W1^T (padded to 96 rows).  Where the contraction runs
    # over HIDDEN neurons (W2, W2^T, W1^T) position 16 ks + 8 hf + e stands for neuron hid(ks, hf, e): the order in which
    # an MFMA accumulator tile holds the previous layer's outputs (include/goslam_neus.h)
    def hid(ks, hf, e):
        return 32 * (ks >> 1) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * hf + (e & 3)
    assert sorted(hid(ks, hf, e) for ks in range(4) for hf in range(2) for e in range(8)) == list(range(64))
    Wv = torch.randn(10240, generator=g)
    P = _pack_mlp_fragments(Wv)
    W1, W2, W3 = Wv[:5120].view(64, 80), Wv[5120:9216].view(64, 64), Wv[9216:].view(16, 64)
    cases = [(0, 0, 0, 0), (1, -1, 63, 7), (1, 0, 33, 2), (0, 1, 47, 5)]
    for base, M, nks, permuted in [(0, W1, 5, False), (10, W2, 4, True), (18, W3.t(), 1, False), (20, W2.t(), 4, True)]:
        for mt, ks, l, e in cases:
            ks = ks % nks
            k = hid(ks, l >> 5, e) if permuted else 16 * ks + 8 * (l >> 5) + e
            assert P[base + mt * nks + ks, l, e] == M[32 * mt + (l & 31), k]
    W1T = torch.zeros(96, 64)
    W1T[:80] = W1.t()
    for mt, ks, l, e in [(0, 0, 0, 0), (2, 3, 63, 7), (2, 1, 15, 3), (2, 2, 16, 0), (1, 1, 40, 6)]:
        assert P[28 + mt * 4 + ks, l, e] == W1T[32 * mt + (l & 31), hid(ks, l >> 5, e)]


class _FakeCorr:
    """stand-in for the HIP CorrBlock on the CPU: tracks only how many edges it holds."""

    def __init__(self, fmap1, fmap2, **kw):
        self.n = fmap1.shape[1]

    def cat(self, other):
        self.n += other.n
        return self

    def __getitem__(self, keep):            # a boolean mask over the edges, or the list of kept edge indices
        if keep.dtype == torch.bool:
            assert keep.numel() == self.n
            self.n = int(keep.sum())
        else:
            assert keep.numel() <= self.n and (keep.numel() == 0 or int(keep.max()) < self.n)
            self.n = int(keep.numel())
        return self


def test_edge_management_matches_reference_golden(monkeypatch):
    """Replay tests/golden/gen_golden.py's scenario (neighbourhood + proximity proposals with NMS, duplicate filter,
    max_factors retirement, filter_edges, rm_keyframe, ageing, clear_edges) on our FactorGraph and compare every
    edge list after every step with what the reference's own FactorGraph produced (src/factor_graph.py:43-197,
    368-450; fixture factor_graph_edges.npz)."""
    import importlib.util
    import os

    import numpy as np

    import go_slam_amd.factor_graph as FG
    here = os.path.dirname(__file__)
    spec = importlib.util.spec_from_file_location("gen_golden", os.path.join(here, "golden", "gen_golden.py"))
    gen = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gen)
    gold = np.load(os.path.join(here, "golden", "factor_graph_edges.npz"))
    dist, conf = torch.from_numpy(gold["dist"]), torch.from_numpy(gold["conf"])
    d2, c2 = gen.graph_script(dist.shape[0])
    assert torch.equal(d2, dist) and torch.equal(c2, conf)          # the scenario is the committed one
    monkeypatch.setattr(FG, "CorrBlock", _FakeCorr)
    T, h, w = dist.shape[0], 4, 4
    B = T + 2
    video = types.SimpleNamespace(
        ht=h, wd=w, stereo=False, counter=0, disps=torch.ones(B, h, w), poses=torch.from_numpy(gold["poses_in"]).clone(),
        nets=torch.rand(B, 4, h, w), inps=torch.rand(B, 4, h, w), fmaps=torch.rand(B, 1, 4, h, w),
        distance=lambda ii, jj, beta=0.3: dist[ii, jj].clone(),
        reproject=lambda ii, jj: (torch.zeros(1, len(ii), h, w, 2), torch.ones(1, len(ii), h, w, 1)))
    g = FactorGraph(video, update_op=None, device="cpu", corr_impl="volume", max_factors=40, channels_last=False)
    seen = []

    def snapshot(tag):
        seen.append(tag)
        for k in ("ii", "jj", "age", "ii_inac", "jj_inac", "ii_bad", "jj_bad"):
            assert np.array_equal(getattr(g, k).numpy(), gold[f"{tag}_{k}"]), (tag, k)
        E = g.ii.numel()
        assert g.target.shape[1] == E and g.weight.shape[1] == E and g.target_inac.shape[1] == g.ii_inac.numel()
        assert (g.corr.n if g.corr is not None else 0) == E
        if g.net is not None:
            assert g.net.shape[1] == E and g.inp.shape[1] == E
    gen.run_graph_scenario(g, video, dist, conf, lambda n: setattr(video, "counter", n), snapshot)
    assert seen == list(gold["tags"])
    assert np.array_equal(video.poses.numpy(), gold["poses_out"])   # rm_keyframe moved slot ix+1 down


def test_proximity_proposal_corner_cases():
    """propose_proximity_edges: max_factors = -1 stops after the local window (the reference's `len(es) >
    max_factors` test, src/factor_graph.py:436), stereo adds the (i, i) edges, suppressed candidates are skipped."""
    import numpy as np
    d = np.full((4, 6), 5.0, dtype=np.float32)
    es = FactorGraph.propose_proximity_edges(d.copy(), [], 2, 0, 6, 1, 1, 16.0, -1, False)
    assert es == [(2, 1), (1, 2), (3, 2), (2, 3), (4, 3), (3, 4), (5, 4), (4, 5)]
    es = FactorGraph.propose_proximity_edges(d.copy(), [], 2, 0, 6, 1, 1, 16.0, -1, True)
    assert es[:3] == [(2, 2), (2, 1), (1, 2)] and (5, 5) in es
    d2 = d.copy()
    d2[3, 0] = 1.0                                                  # (5, 0): the only candidate left after NMS
    es = FactorGraph.propose_proximity_edges(d2, [(3, 1)], 2, 0, 6, 1, 1, 16.0, 100, False)
    assert es[8:10] == [(5, 0), (0, 5)]
    assert all(e != (3, 1) and e != (2, 0) for e in es[8:])         # existing edge and its NMS neighbourhood


def test_lowmem_index_chunks_and_cache():
    """_lowmem_index: 13-keyframe chunks by source keyframe (empty chunks skipped), rig-strided alt-corr indices with
    the right view for stereo pairs, per-chunk unique sources, BA window / damping rows as the reference derives them
    (src/factor_graph.py:262-300), cached per edge-list version."""
    g = _graph()
    g.ii = torch.tensor([2, 2, 3, 16, 16, 40, 40, 67])                 # no source keyframe in [41, 54) and [54, 67)
    g.jj = torch.tensor([3, 2, 2, 15, 17, 39, 40, 40])
    c = g._lowmem_index(None, None, 2)
    assert (c["t0"], c["t1"]) == (3, 68)
    sels = [ck["sel"].tolist() for ck in c["chunks"]]
    assert sels == [[0, 1, 2], [3, 4], [5, 6], [7]]                     # sources in [2,15) [15,28) [28,41) [41,54)
    first = c["chunks"][0]
    assert first["corr_ii"].tolist() == [4, 4, 6] and first["corr_jj"].tolist() == [6, 5, 4]    # (2,2) is a stereo pair
    assert [ck["uniq"].tolist() for ck in c["chunks"]] == [[2, 3], [16], [40], [67]]
    want = torch.unique(torch.cat([torch.arange(3, 68), g.ii]))
    assert torch.equal(c["damping_index"], want)
    assert g._lowmem_index(None, None, 2) is c
    assert g._lowmem_index(None, None, 1) is not c                     # different rig
    d = g._lowmem_index(5, 30, 2)
    assert (d["t0"], d["t1"]) == (5, 30)
    g.ii = g.ii.clone()                                                # replaced tensor -> new index
    assert g._lowmem_index(5, 30, 2) is not d
