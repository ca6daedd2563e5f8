"""world_size-2 gloo tests (CPU) of the N>1 path: image sharding, job-throughput reduction,
joint-mode gradient all-reduce and whole-batch depth mean."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import gan2shape_amd  # noqa: F401
from gan2shape_amd import sharding


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    r, w, lr = sharding.init_distributed("gloo")
    assert (r, w, lr) == (rank, world, rank)
    out = {}
    # instance mode: round-robin images, every image exactly once over the job
    mine = sharding.shard_indices(7, rank, world)
    gathered = [None] * world
    dist.all_gather_object(gathered, mine)
    out["shards"] = gathered
    # throughput: sum of units / slowest rank
    out["rate"] = sharding.job_throughput(units_local=10 * (rank + 1), seconds_local=1.0 + rank)
    # joint mode: averaged shard gradients == full-batch gradient
    torch.manual_seed(0)
    net = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.Tanh(), torch.nn.Linear(5, 1))
    x = torch.arange(24, dtype=torch.float32).view(4, 6) / 10
    full = net(x).pow(2).mean()
    gfull = torch.autograd.grad(full, list(net.parameters()))
    xs = x[rank * 2:(rank + 1) * 2]
    net(xs).pow(2).mean().backward()
    sharding.allreduce_mean_gradients(net.parameters())
    out["grad_err"] = max(float((p.grad - g).abs().max()) for p, g in zip(net.parameters(), gfull))
    # whole-batch mean (model.py:338) under data parallelism
    out["mean_err"] = float((sharding.global_mean(xs) - x.mean()).abs())
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_two_rank_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=100) for _ in range(world))
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    for r in range(world):
        o = res[r]
        assert o["shards"] == [[0, 2, 4, 6], [1, 3, 5]]
        rate, tmax = o["rate"]
        assert tmax == 2.0 and rate == 30 / 2.0
        assert o["grad_err"] < 1e-6 and o["mean_err"] < 1e-6


def test_single_process_is_a_noop():
    assert sharding.job_throughput(8, 2.0) == (4.0, 2.0)
    p = torch.nn.Parameter(torch.ones(3))
    p.grad = torch.full((3,), 2.0)
    sharding.allreduce_mean_gradients([p])
    assert torch.equal(p.grad, torch.full((3,), 2.0))
    assert sharding.global_mean(torch.tensor([1.0, 3.0])) == 2.0
