"""The `tinycudann` drop-in (go_slam_amd/neus/tcnn_compat.py) under the reference's OWN autograd use.

SURVEY 8b Boundary 2: `tcnn.Encoding` must be differentiable w.r.t. params and x including double backward,
`tcnn.Network` w.r.t. params and x; both must survive share_memory / deepcopy / state_dict / .to(device).
The reference (src/InstantNeuS.py:121-159, 295-370) wraps EVERY forward in torch.enable_grad(), calls
pts.requires_grad_(True), takes autograd.grad(sdf, pts, create_graph=True) and later back-propagates the mapper's
loss through that gradient.  `_RefStyleNeuS` below restates that call sequence on top of the drop-in modules (the
reference tree is not available on the GPU box); gradients are compared with torch.autograd on the differentiable
CPU oracle (oracle/neus_autograd.py), relative L2 < 5e-3 as for the fused pipeline.
"""
import copy

import pytest
import torch
import torch.nn as nn
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def T(built_lib):
    from go_slam_amd.neus import tcnn_compat
    return tcnn_compat


@pytest.fixture(scope="module")
def O():
    from oracle import neus_oracle
    return neus_oracle


def _rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-12))


def _points(n, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.rand(n, 3, generator=g)
    x[: n // 4] = x[: n // 4] * 0.02 + 0.4        # a clump inside a few coarse cells (merged-scatter runs)
    return x


# ------------------------------------------------------------------------------------------------------------------
# tcnn.Encoding: first and second order against autograd on the oracle
# ------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.float16])
def test_encoding_autograd_first_and_second_order(T, O, dev, grad_dtype):
    from oracle import neus_autograd as NA
    meta = O.grid_meta()
    n = 777
    g = torch.Generator().manual_seed(5)
    grid = (torch.rand(int(meta["total"]) * 2, generator=g) - 0.5) * 0.6
    x = _points(n, 6)
    A = torch.randn(n, 32, generator=g)            # weights of the value path
    c = torch.randn(32, generator=g)               # upstream of the autograd.grad call (the reference: W[0,3:])
    B = torch.randn(n, 3, generator=g)             # weights of the gradient path
    if grad_dtype == torch.float16:
        # tcnn's own accumulation mode: fp16 table gradient under a loss scale of 128.  d w / d x reaches the level
        # scale (4096 at the finest level), so O(1) upstream gradients overflow fp16 there exactly as they do in
        # tiny-cuda-nn; realistic loss gradients (per-ray means) are orders of magnitude smaller.
        A, B = A * 0.05, B * 1e-3

    # ---- oracle: everything through torch.autograd on the differentiable restatement
    go = grid.clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    co = c.clone().requires_grad_(True)
    enc_o, dydx_o = NA.grid_encode_diff(xo, go, meta, x_differentiable=True)
    c16 = NA._ste_half(co)
    gx_o = torch.einsum("ncd,c->nd", dydx_o, c16)
    L_o = (enc_o * A).sum() + (gx_o * B).sum()
    L_o.backward()

    # ---- drop-in: the reference's call pattern
    enc_m = T.Encoding(3, dict(otype="HashGrid", n_levels=16, n_features_per_level=2, log2_hashmap_size=19,
                               base_resolution=16, per_level_scale=1.447269237440378)).to(dev)
    enc_m.grad_dtype = grad_dtype
    with torch.no_grad():
        enc_m.params.copy_(grid)
    xg = x.to(dev).requires_grad_(True)
    cg = c.to(dev).requires_grad_(True)
    with torch.enable_grad():
        y = enc_m(xg)
        assert y.dtype == torch.float16 and y.requires_grad
        s = (y.float() * cg).sum(-1, keepdim=True)
        gx = torch.autograd.grad(outputs=s, inputs=xg, grad_outputs=torch.ones_like(s), create_graph=True,
                                 retain_graph=True, only_inputs=True)[0]
    assert gx.requires_grad
    torch.testing.assert_close(y.detach().float().cpu(), enc_o.detach(), rtol=0, atol=1e-3)
    # first-order input gradient (the fp16 cast of the upstream gradient is part of tcnn's contract)
    torch.testing.assert_close(gx.detach().cpu(), gx_o.detach(), rtol=2e-3, atol=2e-3 * float(gx_o.abs().max()))
    L = (y.float() * A.to(dev)).sum() + (gx * B.to(dev)).sum()
    L.backward()
    rep = {"grid": _rel(enc_m.params.grad.cpu(), go.grad), "c": _rel(cg.grad.cpu(), co.grad),
           "x": _rel(xg.grad.cpu(), xo.grad)}
    assert rep["grid"] < 5e-3 and rep["c"] < 5e-3 and rep["x"] < 5e-3, rep


def test_encoding_backward_without_create_graph_and_no_grad_path(T, O, dev):
    """Plain first-order use (loss.backward() only) and torch.no_grad() inference."""
    from oracle import neus_autograd as NA
    meta = O.grid_meta()
    g = torch.Generator().manual_seed(15)
    grid = (torch.rand(int(meta["total"]) * 2, generator=g) - 0.5) * 0.4
    x = _points(300, 16)
    A = torch.randn(300, 32, generator=g)
    go = grid.clone().requires_grad_(True)
    enc_o, _ = NA.grid_encode_diff(x, go, meta)
    (enc_o * A).sum().backward()
    m = T.Encoding(3, {}).to(dev)
    with torch.no_grad():
        m.params.copy_(grid)
        y0 = m(x.to(dev))
    assert not y0.requires_grad
    y = m(x.to(dev))
    (y.float() * A.to(dev)).sum().backward()
    assert _rel(m.params.grad.cpu(), go.grad) < 5e-3
    torch.testing.assert_close(y0, y.detach())


# ------------------------------------------------------------------------------------------------------------------
# tcnn.Network
# ------------------------------------------------------------------------------------------------------------------
def test_network_autograd_matches_oracle(T, O, dev):
    from oracle import neus_autograd as NA
    g = torch.Generator().manual_seed(25)
    n = 1000
    params = (torch.rand(O.mlp_num_params(), generator=g) - 0.5) * 0.5
    x = torch.randn(n, 67, generator=g) * 0.7
    A = torch.randn(n, 3, generator=g)
    po = params.clone().requires_grad_(True)
    xo = x.clone().requires_grad_(True)
    y_o = NA.mlp_diff(xo, po)
    (torch.sigmoid(y_o) * A).sum().backward()
    net = T.Network(67, 3, dict(otype="FullyFusedMLP", activation="ReLU", output_activation="none", n_neurons=64,
                               n_hidden_layers=2)).to(dev)
    with torch.no_grad():
        net.params.copy_(params)
    xg = x.to(dev).requires_grad_(True)
    y = net(xg)
    assert y.dtype == torch.float16 and tuple(y.shape) == (n, 3)
    torch.testing.assert_close(y.detach().float().cpu(), y_o.detach(), rtol=5e-3, atol=4e-3)
    (torch.sigmoid(y.float()) * A.to(dev)).sum().backward()
    rep = {"params": _rel(net.params.grad.cpu(), po.grad), "x": _rel(xg.grad.cpu(), xo.grad)}
    assert rep["params"] < 5e-3 and rep["x"] < 5e-3, rep


# ------------------------------------------------------------------------------------------------------------------
# the reference's InstantNeuS call sequence on the drop-in
# ------------------------------------------------------------------------------------------------------------------
class _RefStyleNeuS(nn.Module):
    """Call-for-call restatement of the reference model on `tcnn`-style modules: xyz + hash encoding -> Linear;
    SDF gradient by autograd.grad(create_graph=True) under enable_grad on a fresh leaf; boolean-mask scatter;
    NeuS alpha; colour MLP on [sin(pB) | normal | feat]; cumprod compositing (src/InstantNeuS.py:35-370)."""

    def __init__(self, tcnn, bound, device):
        super().__init__()
        self.register_buffer("bound", torch.tensor(bound).float())
        self.register_buffer("realtime_bound", torch.tensor(bound).float())
        with torch.cuda.device(device):
            self.grid = tcnn.Encoding(n_input_dims=3, encoding_config={
                "otype": "HashGrid", "n_levels": 16, "n_features_per_level": 2, "log2_hashmap_size": 19,
                "base_resolution": 16, "per_level_scale": 1.447269237440378, "include_xyz": True})
            self.mlp = tcnn.Network(n_input_dims=33 + 3 + 31, n_output_dims=3, network_config={
                "otype": "FullyFusedMLP", "activation": "ReLU", "output_activation": "none", "n_neurons": 64,
                "n_hidden_layers": 2})
        self.sdf_layer = nn.Linear(3 + self.grid.n_output_dims, 32)
        self.color_B = nn.Parameter(torch.randn(3, 33) * 25.0)
        self.variance = nn.Parameter(torch.tensor(0.2))

    def sdf_feat(self, pts):
        b = self.bound
        p = ((pts - b[:, 0]) / (b[:, 1] - b[:, 0]) * 2.0 - 1.0).clamp(min=-1.0, max=1.0)
        h = torch.cat([p, self.grid((p + 1) / 2)], dim=-1)           # fp16 features promoted by cat
        out = self.sdf_layer(h)
        return out[:, 0:1], out[:, 1:]

    def sdf_with_gradient(self, pts):
        with torch.enable_grad():
            pts.requires_grad_(True)
            sdf, feat = self.sdf_feat(pts)
            ones = torch.ones_like(sdf, requires_grad=False)
            grad = torch.autograd.grad(outputs=sdf, inputs=pts, grad_outputs=ones, create_graph=True,
                                       retain_graph=True, only_inputs=True)[0]
        return sdf, feat, grad

    def alpha(self, sdf, grads, dirs, dists):
        inv_s = (torch.ones_like(sdf) * torch.exp(self.variance * 10.0)).clip(1e-6, 1e6)
        cos = (dirs * grads).sum(1, keepdim=True)
        it = -F.relu(-cos)                                            # cos_anneal_ratio = 1
        nxt, prv = sdf + it * dists.reshape(-1, 1) / 2.0, sdf - it * dists.reshape(-1, 1) / 2.0
        pc, nc = torch.sigmoid(prv * inv_s), torch.sigmoid(nxt * inv_s)
        return ((pc - nc + 1e-5) / (pc + 1e-5)).clip(0.0, 1.0)

    def forward(self, rays_o, rays_d, z_vals, dists):
        n, s = z_vals.shape
        dev = z_vals.device
        zm = z_vals + dists / 2.0
        pts = (rays_o[:, None] + rays_d[:, None] * zm[..., None]).reshape(-1, 3)
        dirs = rays_d[:, None].expand(n, s, 3).reshape(-1, 3)
        rb = self.realtime_bound
        mask = ((pts < rb[:, 1]) & (pts > rb[:, 0])).all(-1)
        if mask.float().sum() < 1:
            mask[:100] = True
        o_sdf, o_feat, o_grad = self.sdf_with_gradient(pts[mask])
        sdf = torch.ones(n * s, 1, device=dev) * 100
        feat = torch.zeros(n * s, o_feat.shape[1], device=dev, dtype=o_feat.dtype)
        grads = torch.zeros(n * s, 3, device=dev, dtype=o_grad.dtype)
        sdf[mask], feat[mask], grads[mask] = o_sdf, o_feat, o_grad
        alpha = self.alpha(sdf, grads, dirs, dists)
        emb = torch.sin(pts[mask] @ self.color_B)
        o_rgb = torch.sigmoid(self.mlp(torch.cat([emb, grads[mask], feat[mask]], dim=1)))
        rgb = torch.zeros(n * s, 3, device=dev, dtype=o_rgb.dtype)
        rgb[mask] = o_rgb
        sdf, rgb = sdf.reshape(n, s), rgb.reshape(n, s, 3)
        alpha = (alpha * mask[:, None]).reshape(n, s)
        grads, m2 = grads.reshape(n, s, 3), mask.reshape(n, s)
        w = alpha * torch.cumprod(torch.cat([torch.ones(n, 1, device=dev), 1 - alpha + 1e-7], 1), 1)[:, :-1]
        depth = (zm * w).sum(1, keepdim=True)
        gerr = ((torch.linalg.norm(grads, ord=2, dim=2) - 1.0) ** 2 * m2).mean().unsqueeze(0)
        return {"color": (rgb * w[..., None]).sum(1), "depth": depth,
                "depth_variance": ((zm - depth) ** 2 * w).sum(1, keepdim=True),
                "normal": (grads * w[..., None] * m2[..., None]).sum(1), "weight_sum": w.sum(1, keepdim=True),
                "sdf": sdf, "z_vals": zm, "gradient_error": gerr}


def _rays(n, seed):
    g = torch.Generator().manual_seed(seed)
    o = torch.rand(n, 3, generator=g) * 4 - 2
    d = F.normalize(torch.randn(n, 3, generator=g), dim=1)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[torch.rand(n, generator=g) < 0.1] = 0
    return o, d, gt


@pytest.mark.parametrize("grad_dtype", [torch.float32, torch.float16])
def test_reference_call_sequence_runs_on_dropin_and_matches_oracle_gradients(T, O, dev, grad_dtype):
    from oracle import neus_autograd as NA
    P = O.make_params(21, grid_init=0.3, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
    P["rt_bound"] = torch.tensor([[-2.2, 2.3], [-2.4, 2.1], [-2.0, 2.2]])
    o, d, gt = _rays(48, 22)
    g = torch.Generator().manual_seed(23)
    z, dist = O.render_sample(o, d, gt, P["bound"], 24, 48, torch.rand(24, generator=g))
    col = torch.rand(48, 3, generator=g)
    Pd = {k: (v.clone().requires_grad_(True) if k in ("grid", "sdf_w", "sdf_b", "color_B", "mlp") else v)
          for k, v in P.items()}
    Pd["variance"] = torch.tensor(0.2, requires_grad=True)
    ref_out = NA.neus_forward_diff(o, d, z, dist, Pd)
    ref_loss = NA.mapping_loss(ref_out, col, gt)
    ref_loss.backward()

    model = _RefStyleNeuS(T, P["bound"].tolist(), dev).to(dev)
    model.grid.grad_dtype = grad_dtype
    with torch.no_grad():
        model.grid.params.copy_(P["grid"])
        model.sdf_layer.weight.copy_(P["sdf_w"])
        model.sdf_layer.bias.copy_(P["sdf_b"])
        model.color_B.copy_(P["color_B"])
        model.mlp.params.copy_(P["mlp"])
        model.variance.fill_(0.2)
        model.realtime_bound.copy_(P["rt_bound"])
    # inference exactly as the reference renders: under no_grad the model still differentiates internally
    with torch.no_grad():
        out0 = model(o.to(dev), d.to(dev), z.to(dev), dist.to(dev))
    torch.testing.assert_close(out0["sdf"].cpu(), ref_out["sdf"].detach(), rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(out0["color"].cpu(), ref_out["color"].detach(), rtol=0, atol=5e-3)
    torch.testing.assert_close(out0["normal"].cpu(), ref_out["normal"].detach(), rtol=5e-3, atol=5e-3)
    # training step
    out = model(o.to(dev), d.to(dev), z.to(dev), dist.to(dev))
    loss = NA.mapping_loss(out, col.to(dev), gt.to(dev))
    loss.backward()
    torch.testing.assert_close(loss.detach().cpu(), ref_loss.detach(), rtol=3e-3, atol=2e-4)
    pairs = {"grid": (model.grid.params.grad, Pd["grid"].grad), "sdf_w": (model.sdf_layer.weight.grad, Pd["sdf_w"].grad),
             "sdf_b": (model.sdf_layer.bias.grad, Pd["sdf_b"].grad), "color_B": (model.color_B.grad, Pd["color_B"].grad),
             "mlp": (model.mlp.params.grad, Pd["mlp"].grad),
             "variance": (model.variance.grad.reshape(1), Pd["variance"].grad.reshape(1))}
    report = {k: _rel(a.cpu().float(), b) for k, (a, b) in pairs.items()}
    assert all(r < 5e-3 for r in report.values()), report


def test_dropin_equals_fused_pipeline_forward(T, O, dev, built_lib):
    """The reference-style model on the drop-in and the package's fused InstantNeuS render the same rays."""
    import go_slam_amd.neus as neus
    P = O.make_params(41, grid_init=0.3, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
    o, d, gt = _rays(96, 42)
    z, dist = O.render_sample(o, d, gt, P["bound"], 24, 48, None)
    ref = _RefStyleNeuS(T, P["bound"].tolist(), dev).to(dev)
    fused = neus.InstantNeuS({}, P["bound"].tolist()).to(dev)
    with torch.no_grad():
        for dst, key in ((ref.grid.params, "grid"), (ref.sdf_layer.weight, "sdf_w"), (ref.sdf_layer.bias, "sdf_b"),
                         (ref.color_B, "color_B"), (ref.mlp.params, "mlp"),
                         (fused.sdf_network.encoding.encoding.params, "grid"), (fused.sdf_network.sdf_layer.weight, "sdf_w"),
                         (fused.sdf_network.sdf_layer.bias, "sdf_b"), (fused.color_network._B, "color_B"),
                         (fused.color_network.network.params, "mlp")):
            dst.copy_(P[key])
        fused.variance_network.variance.fill_(0.2)
        a = ref(o.to(dev), d.to(dev), z.to(dev), dist.to(dev))
        b = fused(o.to(dev), d.to(dev), z.to(dev), dist.to(dev))
    torch.testing.assert_close(a["sdf"], b["sdf"], rtol=1e-3, atol=2e-4)
    torch.testing.assert_close(a["depth"], b["depth"], rtol=2e-3, atol=2e-3)
    torch.testing.assert_close(a["color"], b["color"], rtol=0, atol=6e-3)


def test_dropin_module_contract(T, dev):
    """share_memory / deepcopy / state_dict / .to(device) (slam.py:122, mesher.py:247) and cache invalidation."""
    enc = T.Encoding(3, {})
    net = T.Network(67, 3, {})
    assert enc.n_output_dims == 32 and [n for n, _ in enc.named_parameters()] == ["params"]
    assert [n for n, _ in net.named_parameters()] == ["params"] and net.params.numel() == 10240
    enc = enc.to(dev)
    enc.share_memory()
    x = torch.rand(100, 3, device=dev)
    with torch.no_grad():
        enc.params.uniform_(-0.5, 0.5)
        y1 = enc(x)
        twin = copy.deepcopy(enc)
        torch.testing.assert_close(twin(x), y1)
        fresh = T.Encoding(3, {}).to(dev)
        fresh.load_state_dict(enc.state_dict())
        torch.testing.assert_close(fresh(x), y1)
    # an optimiser step must invalidate the cached fp16 table
    opt = torch.optim.SGD(enc.parameters(), lr=10.0)
    enc(x).float().sum().backward()
    opt.step()
    with torch.no_grad():
        y2 = enc(x)
    assert not torch.equal(y1, y2)
    torch.testing.assert_close(twin(x), y1)       # the copy kept its own parameters


def test_fused_adamw_needs_the_version_hook(T, dev):
    """torch.optim.AdamW(fused=True) leaves `_version` untouched (so a version-keyed fp16 cache would go stale);
    optimisers from go_slam_amd.neus.mapper.make_optimizer advance it in a step hook."""
    import go_slam_amd.neus as neus
    from go_slam_amd.neus.mapper import make_optimizer
    model = neus.InstantNeuS({}, [[-1.0, 1.0]] * 3).to(dev)
    enc = model.sdf_network.encoding.encoding
    x = torch.rand(64, 3, device=dev)
    with torch.no_grad():
        enc.params.uniform_(-0.5, 0.5)
        y1 = enc(x)
    opt = make_optimizer(model)
    v0 = enc.params._version
    enc(x).float().sum().backward()
    for p in model.parameters():
        if p.grad is None:
            p.grad = torch.zeros_like(p)
    opt.step()
    assert enc.params._version > v0
    with torch.no_grad():
        y2 = enc(x)
    assert not torch.equal(y1, y2), "the encoding still serves the pre-step fp16 table"
