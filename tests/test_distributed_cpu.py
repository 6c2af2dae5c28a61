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
