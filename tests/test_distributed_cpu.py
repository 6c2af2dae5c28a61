"""CPU, gloo, world_size 2: the sharded mapping step's loss normalisation + flat-gradient
all-reduce reproduce the single-process gradient.  The compute engine on the CPU is the
differentiable oracle (test infrastructure); the code under test is go_slam_amd/neus/distributed.py."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _problem():
    from oracle import neus_oracle as NO
    P = NO.make_params(41, grid_init=0.3, bound=((-2.5, 2.5), (-2.5, 2.5), (-2.5, 2.5)))
    g = torch.Generator().manual_seed(42)
    n = 22                                   # odd split: 11 / 11 valid counts differ per shard
    o = torch.rand(n, 3, generator=g) * 4 - 2
    d = torch.nn.functional.normalize(torch.randn(n, 3, generator=g), dim=1)
    gt = torch.rand(n, generator=g) * 3.5 + 0.5
    gt[[1, 2, 3, 15]] = 0                    # 3 invalid rays in shard 0, 1 in shard 1
    col = torch.rand(n, 3, generator=g)
    z, dist_ = NO.render_sample(o, d, gt, P["bound"], 8, 16, torch.rand(8, generator=g))
    return P, o, d, gt, col, z, dist_


def _grads(P, o, d, gt, col, z, dist_, group, rank, world):
    from go_slam_amd.neus.distributed import FlatGradReducer, mapping_loss_sharded, shard_rays
    from oracle import neus_autograd as NA, neus_oracle as NO
    names = ("grid", "sdf_w", "sdf_b", "color_B", "mlp")
    Pd = {k: (v.clone().requires_grad_(True) if k in names else v) for k, v in P.items()}
    Pd["variance"] = torch.tensor(0.2, requires_grad=True)
    if world > 1:
        o, d, gt, col, z, dist_ = shard_rays([o, d, gt, col, z, dist_], rank, world)
    ret = NA.neus_forward_diff(o, d, z, dist_, Pd)
    sdf_err = lambda s, zz, g_: NO.compute_sdf_error(s, zz, g_, 0.16, 5)
    loss, glob = mapping_loss_sharded(ret, col, gt, sdf_err, group)
    loss.backward()
    params = [Pd[k] for k in names] + [Pd["variance"]]
    if world > 1:
        FlatGradReducer(params).reduce(group)
    return glob, [p.grad.clone() for p in params]


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    glob, grads = _grads(*_problem(), None, rank, world)
    if rank == 0:
        torch.save({"loss": glob, "grads": grads}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    from go_slam_amd.neus.distributed import shard_bounds
    for n in (0, 1, 7, 4096, 4400, 32768):
        for w in (1, 2, 3, 8):
            spans = [shard_bounds(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_two_rank_gradient_equals_single_process(tmp_path):
    ref_loss, ref_grads = _grads(*_problem(), None, 0, 1)
    out = str(tmp_path / "r0.pt")
    port = 29500 + (os.getpid() % 2000)
    mp.start_processes(_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = torch.load(out)
    assert abs(got["loss"] - ref_loss) < 1e-5 * max(1.0, abs(ref_loss))
    for a, b in zip(got["grads"], ref_grads):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-6)


def test_bench_gpus_flag_spawns_that_many_ranks():
    """`python bench.py --gpus 2` with no launcher in the environment must start 2 ranks itself (the driver's N > 1
    command shape is `torch.distributed.run ... bench.py --gpus N`, but a plain call must not silently measure one
    GPU).  GS_BENCH_LAUNCH_ONLY=1 stops after the rendezvous, so this runs without a GPU."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["GS_BENCH_LAUNCH_ONLY"] = "1"
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["rccl_ranks"] == 2
    # and under an external launcher the script must NOT nest another one
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    res = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1"], env=env,
                         stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
    assert json.loads([l for l in res.stdout.splitlines() if l.startswith("{")][0])["n_gpus"] == 1


# ---- sharded optimiser (FlatAdamW with world > 1): reduce-scatter -> slice step -> all-gather -----------------------
class _TorchOptKernels:
    """torch restatement of csrc/map_opt.hip's two kernels (TEST infrastructure: lets the gloo test drive the sharding
    logic -- slices, padding, collectives, master sync -- on CPU tensors; the product only ever passes HipOptKernels)."""

    def sqnorm(self, out, g16, inv_scale16, g32):
        out += (g16.float() * inv_scale16).double().pow(2).sum().float()
        if g32 is not None:
            out += g32.double().pow(2).sum().float()

    def adamw(self, p, m, v, p16, g16, inv_scale16, pd, md, vd, p16d, g32, h, step, step_dev, sqnorm):
        t = int(step_dev.item())
        coef = min(1.0, h["max_norm"] / (float(sqnorm.sqrt()) + 1e-6))
        for P, M, V, P16, g, lr in ((p, m, v, p16, g16.float() * (inv_scale16 * coef), h["lr16"]),
                                    (pd, md, vd, p16d, g32 * coef, h["lr32"])):
            P.mul_(1 - lr * h["wd"])
            M.mul_(h["b1"]).add_(g, alpha=1 - h["b1"])
            V.mul_(h["b2"]).addcmul_(g, g, value=1 - h["b2"])
            denom = V.sqrt() / (1 - h["b2"] ** t) ** 0.5 + h["eps"]
            P.addcdiv_(M, denom, value=-lr / (1 - h["b1"] ** t))
            P16.copy_(P.half())


class _TinyModel(torch.nn.Module):
    """just the attribute tree FlatAdamW binds (table of 1000 entries: NOT divisible by 8 * world -> padded slices)"""

    def __init__(self):
        super().__init__()
        from go_slam_amd.neus.tcnn_compat import _HalfCache
        torch.manual_seed(4)                      # nn.Linear draws from the global generator
        g = torch.Generator().manual_seed(5)
        mk = lambda *s: torch.nn.Parameter(torch.randn(*s, generator=g) * 0.1)
        enc = torch.nn.Module(); enc.params = mk(1000); enc._half = _HalfCache()
        outer = torch.nn.Module(); outer.encoding = enc
        self.sdf_network = torch.nn.Module()
        self.sdf_network.encoding = outer
        self.sdf_network.sdf_layer = torch.nn.Linear(35, 32)
        self.color_network = torch.nn.Module()
        net = torch.nn.Module(); net.params = mk(10240); net._half = _HalfCache()
        self.color_network.network = net
        self.color_network._B = mk(3, 33)
        self.variance_network = torch.nn.Module()
        self.variance_network.variance = torch.nn.Parameter(torch.tensor(0.2))


def _opt_run(rank, world, steps=3):
    """`steps` optimiser steps on synthetic gradients; rank r contributes share r of every gradient"""
    from go_slam_amd.neus.mapper import FlatAdamW
    model = _TinyModel()
    flat = FlatAdamW(model, rank=rank, world=world, kernels=_TorchOptKernels())
    g = torch.Generator().manual_seed(6)
    for _ in range(steps):
        # integer-valued fp16 shares: their fp16 sum is exact in any order, so sharded == single-process bit for bit
        shares16 = [torch.randint(-40, 40, (1000,), generator=g).half() for _ in range(2)]
        shares32 = [torch.randn(flat.nd, generator=g) for _ in range(2)]
        tab = flat.grad_table()
        if world == 1:
            tab.copy_(shares16[0] + shares16[1])
            flat.g32[:flat.nd] = shares32[0] + shares32[1]
        else:
            tab.copy_(shares16[rank])
            flat.g32[:flat.nd] = shares32[rank]
        flat.g32[flat.nd] = float(rank + 1)           # "loss" share: summed by the dense all-reduce
        flat.step(1.0 / 128.0)
    loss = float(flat.g32[flat.nd])
    if world > 1:
        # the all-gather of the updated fp16 table is DEFERRED: still in flight after step(), completed by its next reader
        enc = model.sdf_network.encoding.encoding
        assert flat._gather_wait is not None and enc._half._pending is not None
        enc._half.get(enc.params)
        assert flat._gather_wait is None and enc._half._pending is None
    stale = flat.P[:flat.n16].clone()
    flat.sync_master()
    named = {k: flat.P[a:b].clone() for k, (a, b) in flat.slices.items()}
    named16 = {k: flat.P16[a:b].clone() for k, (a, b) in flat.slices.items()}
    assert model.sdf_network.encoding.encoding.params.data_ptr() == flat.P.data_ptr()
    flat.check_bindings()
    return dict(loss=loss, named=named, named16=named16, stale=stale, pad=(flat.n16p, flat.slice), steps=flat.steps,
                sent=flat.collective_bytes())


def _opt_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    res = _opt_run(rank, world)
    torch.save(res, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_optimizer_equals_single_process_step(tmp_path):
    """FlatAdamW sharded over 2 gloo ranks (reduce-scatter of the fp16 table gradient, scalar all-reduce for the clip
    norm, AdamW on the owned slice + the replicated dense part, all-gather of the fp16 working copy, fp32 master gathered
    on demand) == the single-process optimiser on the summed gradients, for a table whose size is not a multiple of the
    slice granularity (padding)."""
    ref = _opt_run(0, 1)
    out = str(tmp_path / "opt")
    port = 29700 + (os.getpid() % 2000)
    mp.start_processes(_opt_worker, args=(2, port, out), nprocs=2, join=True, start_method="spawn")
    got = [torch.load(out + f".{r}") for r in range(2)]
    assert got[0]["pad"] == (1008, 504) and ref["pad"] == (1000, 1000)
    assert got[0]["loss"] == got[1]["loss"] == 3.0 and got[0]["sent"] > 0 and ref["sent"] == 0
    for r in range(2):
        for k in ref["named"]:
            torch.testing.assert_close(got[r]["named"][k], ref["named"][k], rtol=1e-6, atol=1e-7, msg=lambda m: f"{k}: {m}")
            assert torch.equal(got[r]["named16"][k], got[0]["named16"][k])
            torch.testing.assert_close(got[r]["named16"][k].float(), ref["named16"][k].float(), rtol=1e-3, atol=1e-6)
    # before sync_master a rank's fp32 master is current only on its own slice
    assert not torch.equal(got[0]["stale"], ref["named"]["grid"]) and torch.equal(got[0]["stale"][:504], got[0]["named"]["grid"][:504])


def _selftest_worker(rank, world, port, out, break_rank):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    import bench
    if break_rank:                  # the collective layer refuses (on every rank: an API / backend refusal is symmetric)
        from go_slam_amd.neus import distributed as D

        def refuse(*a, **k):
            raise RuntimeError("refused")
        D.all_reduce_sum_ = refuse
    ok, err = bench.collectives_selftest(torch.device("cpu"), rank, world)
    torch.save({"ok": ok, "err": err}, out + f".{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("break_rank", [False, True])
def test_bench_collective_selftest_agrees_across_ranks(tmp_path, break_rank):
    """bench.py's guard of its path-M legs under N > 1: the sharded step's collective calls on small tensors, results
    checked against their closed forms; a refusal is caught, agreed on with one MIN all-reduce and costs the path-M legs
    only -- the tracking headline is printed either way.  (A refusal on ONE rank alone would leave its peers inside the
    refused collective: nothing above the collective library can recover that, so it is not what this guards.)"""
    out = str(tmp_path / "st")
    port = 29900 + (os.getpid() % 2000)
    mp.start_processes(_selftest_worker, args=(3, port, out, break_rank), nprocs=3, join=True, start_method="spawn")
    got = [torch.load(out + f".{r}") for r in range(3)]
    if not break_rank:
        assert all(g["ok"] for g in got), got
    else:
        assert not any(g["ok"] for g in got)
        assert all("refused" in g["err"] for g in got)
