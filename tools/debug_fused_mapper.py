import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import go_slam_amd.neus as N
from go_slam_amd.neus.mapper import MapTrainer
from oracle import neus_oracle as O
dev = torch.device("cuda:0")
P = O.make_params(51, grid_init=0.05, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
g = torch.Generator().manual_seed(52)
n = 512
o = torch.rand(n, 3, generator=g) * 4 - 2
d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
gt = torch.rand(n, generator=g) * 3.5 + 0.5
col = torch.rand(n, 3, generator=g); pr = torch.rand(24, generator=g)
args = [t.to(dev) for t in (o, d, col, gt, pr)]
names = ["sdf_network.encoding.encoding.params", "sdf_network.sdf_layer.weight", "sdf_network.sdf_layer.bias",
         "color_network._B", "color_network.network.params", "variance_network.variance"]
res = {}
for fused in (False, True):
    model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
    with torch.no_grad():
        model.sdf_network.encoding.encoding.params.copy_(P["grid"]); model.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
        model.sdf_network.sdf_layer.bias.copy_(P["sdf_b"]); model.color_network._B.copy_(P["color_B"])
        model.color_network.network.params.copy_(P["mlp"]); model.variance_network.variance.fill_(P["variance"])
    before = {k: v.detach().clone() for k, v in model.named_parameters()}
    tr = MapTrainer(model, N.Renderer(N_samples=24, N_surface=48), fused=fused)
    l0 = float(tr.step(*args))
    after = {k: v.detach().clone() for k, v in model.named_parameters()}
    l1 = float(tr.step(*args))
    res[fused] = (before, after)
    print("fused" if fused else "ref  ", "loss0", l0, "loss1", l1)
    for k in names:
        dlt = (after[k] - before[k]).float()
        print(f"   {k:44s} |delta| max {float(dlt.abs().max()):.3e} mean {float(dlt.abs().mean()):.3e} nonzero {int((dlt != 0).sum())}/{dlt.numel()}")
    if fused:
        print("   sqnorm", float(tr.flat.sqnorm), "steps", tr.flat.steps, "P16 vs P max", float((tr.flat.P16.float() - tr.flat.P).abs().max()))
for k in names:
    a, b = res[True][1][k].float(), res[False][1][k].float()
    print(f"{k:44s} fused-vs-ref after 1 step: max {float((a - b).abs().max()):.3e}")

print("---- flat AdamW kernel vs torch formulas at full size")
model = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
with torch.no_grad():
    model.sdf_network.encoding.encoding.params.copy_(P["grid"]); model.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
    model.color_network._B.copy_(P["color_B"]); model.color_network.network.params.copy_(P["mlp"])
tr = MapTrainer(model, N.Renderer(N_samples=24, N_surface=48), fused=True)
F = tr.flat
for step in (1, 2):
    loss, grid16, inv = tr.fused_gradients(*args)
    p0, m0, v0 = F.P.clone(), F.M.clone(), F.V.clone()
    gfull = torch.cat([grid16.float() * inv, F.g32])
    norm = gfull.double().pow(2).sum().sqrt().float()
    coef = torch.clamp(35.0 / (norm + 1e-6), max=1.0)
    g = gfull * coef
    lr = torch.cat([torch.full((F.n16,), 1e-2, device=dev), torch.full((F.n - F.n16,), 1e-3, device=dev)])
    pe = p0 * (1 - lr * 0.01)
    me = 0.9 * m0 + 0.1 * g
    ve = 0.999 * v0 + 0.001 * g * g
    bc1, bc2 = 1 - 0.9 ** step, 1 - 0.999 ** step
    pe = pe - (lr / bc1) * (me / (ve.sqrt() / bc2 ** 0.5 + 1e-8))
    F.step(grid16, inv)
    torch.cuda.synchronize()
    dp = (F.P - pe).abs()
    print(f"step {step}: sqnorm kernel {float(F.sqnorm):.4f} vs {float(norm) ** 2:.4f}; P max diff {float(dp.max()):.3e} (#>1e-6: {int((dp > 1e-6).sum())}); "
          f"M diff {float((F.M - me).abs().max()):.3e} V diff {float((F.V - ve).abs().max()):.3e}; P16-half(P) {float((F.P16.float() - F.P.half().float()).abs().max()):.3e}")
    bad = (dp > 1e-6).nonzero().reshape(-1)
    if bad.numel():
        print("   bad idx:", bad[:10].tolist(), " mod 8:", sorted(set((bad % 8).tolist())), " max idx", int(bad.max()), "n16", F.n16,
              " g at bad:", g[bad[:5]].tolist(), " p0:", p0[bad[:5]].tolist(), " got:", F.P[bad[:5]].tolist(), " want:", pe[bad[:5]].tolist())

print("---- table gradients: autograd path vs no-autograd path, element by element")
from go_slam_amd.neus.distributed import mapping_loss_sharded
def fresh():
    m = N.InstantNeuS({}, P["bound"].tolist()).to(dev)
    with torch.no_grad():
        m.sdf_network.encoding.encoding.params.copy_(P["grid"]); m.sdf_network.sdf_layer.weight.copy_(P["sdf_w"])
        m.color_network._B.copy_(P["color_B"]); m.color_network.network.params.copy_(P["mlp"])
    return m
R = N.Renderer(N_samples=24, N_surface=48)
gs = []
for rep_ in range(2):
    m = fresh()
    z, dd = R.sample(args[0], args[1], m.bound, args[3], args[4])
    la, _ = mapping_loss_sharded(R.eval_points(args[0], args[1], z, dd, m, None), args[2], args[3], m.compute_sdf_error)
    la.backward()
    gs.append(m.sdf_network.encoding.encoding.params.grad.clone())
m = fresh()
tr = MapTrainer(m, R, fused=True)
_, g16, inv = tr.fused_gradients(*args)
gf = g16.float() * inv
_, g16b, _ = tr.fused_gradients(*args)
gf2 = g16b.float() * inv
def cmp(name, a, b):
    d = (a - b).abs()
    nz = (a != 0) | (b != 0)
    flips = ((a > 0) & (b < 0)) | ((a < 0) & (b > 0)) | ((a == 0) != (b == 0))
    print(f"  {name}: nonzero entries {int(nz.sum())}, differing {int((d > 0).sum())}, sign/zero flips {int(flips.sum())}, max |diff| {float(d.max()):.3e}, max |g| {float(a.abs().max()):.3e}")
cmp("autograd run 1 vs run 2", gs[0], gs[1]); cmp("fused run 1 vs run 2", gf, gf2); cmp("autograd vs fused", gs[0], gf)
