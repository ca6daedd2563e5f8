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


def test_grad_bucket_pack_bind_and_inplace_accumulation():
    """sharding.GradBucket: gradients travel into ONE persistent flat buffer (a missing gradient as zeros), the
    parameters' .grad become views of it (nothing copied back), a gradient accumulated in place into a bound view
    is not copied again, and the same parameter list always gets the same bucket."""
    ps = [torch.nn.Parameter(torch.zeros(3, 2)), torch.nn.Parameter(torch.zeros(5)), torch.nn.Parameter(torch.zeros(4))]
    b = sharding.bucket_of(ps)
    assert sharding.bucket_of(ps) is b and b.flat.numel() == 15
    ps[0].grad, ps[2].grad = torch.arange(6.0).view(3, 2), torch.full((4,), 7.0)
    b.flat.fill_(-1.0)
    b.pack()
    assert torch.equal(b.flat, torch.cat([torch.arange(6.0), torch.zeros(5), torch.full((4,), 7.0)]))
    b.bind()
    assert all(p.grad.data_ptr() == v.data_ptr() and p.grad.shape == p.shape for p, v in zip(ps, b.views))
    ps[1].grad.add_(2.0)                       # autograd accumulates in place into a bound view
    b.pack()                                   # nothing to move
    assert torch.equal(b.flat[6:11], torch.full((5,), 2.0))
    ps[1].grad = None                          # zero_grad(set_to_none=True), then a step without this gradient
    b.pack()
    assert torch.equal(b.flat[6:11], torch.zeros(5)) and torch.equal(b.flat[:6], torch.arange(6.0))


# ------------------------------------------------------------------ joint trainer (GeneralizingTrainer2)
class _ToyModel(torch.nn.Module):
    """CPU stand-in with the model step API the trainers drive (trainer.py:40-48,103-104,147): the
    real GAN2Shape needs the GPU library; the trainer's batching / collectives do not."""

    def __init__(self, config, debug=False, device="cpu"):
        super().__init__()
        torch.manual_seed(7)
        for name in ("albedo", "offset_encoder", "lighting", "viewpoint", "depth"):
            setattr(self, f"{name}_net", torch.nn.Linear(12, 4))
        self.batch_mean = None
        self.calls = []
        self.centres = []   # (step, centre used, this process's own mean)

    def _feat(self, images):
        return images.reshape(len(images), -1)[:, :12]

    def depth_net_forward(self, inputs, prior):
        d = self.depth_net(self._feat(inputs))
        mean = d.mean() if self.batch_mean is None else self.batch_mean(d)
        return ((d - mean) ** 2).mean() + 0 * prior.sum(), d

    def forward_step1(self, images, latents, collected, **kw):
        self.calls.append((1, len(images)))
        a = self.albedo_net(self._feat(images))
        d = self.depth_net(self._feat(images)).detach()
        mean = d.mean() if self.batch_mean is None else self.batch_mean(d)
        self.centres.append((1, float(mean), float(d.mean())))
        loss = ((a - mean) ** 2).mean()
        z = torch.zeros(len(images), 1)
        return loss, (a, z, z, a, d, None if len(images) == 1 else [None] * len(images))

    def forward_step2(self, image, latent, collected, n_proj_samples=8, **kw):
        self.calls.append((2, len(image)))
        assert collected[0].shape[0] == 1 and not collected[0].requires_grad
        e = self.offset_encoder_net(self._feat(image))
        return (e ** 2).mean() + collected[0].sum() * 0, (e.detach(), e.detach())

    def forward_step3(self, image, latent, collected, **kw):
        self.calls.append((3, len(image)))
        # the inner step-1 pass of step 3 centres the depth (model.py:236,338)
        d = self.depth_net(self._feat(image)).detach()
        mean = d.mean() if self.batch_mean is None else self.batch_mean(d)
        self.centres.append((3, float(mean), float(d.mean())))
        out = sum(getattr(self, f"{n}_net")(self._feat(image)).mean() for n in ("lighting", "viewpoint", "depth", "albedo"))
        return (out - collected[0].mean()) ** 2, None


def _toy_data(n=4):
    g = torch.Generator().manual_seed(3)
    return [(torch.randn(3, 16, 16, generator=g), torch.randn(8, generator=g), i) for i in range(n)]


_TOY_CFG = {"image_size": 16, "category": "face", "n_proj_samples": 2, "n_epochs_prior": 2,
            "n_epochs_generalized": 2, "prior_name": "ellipsoid"}


def _joint_worker(rank, world, port, q, stages):
    from gan2shape_amd.trainer import GeneralizingTrainer2
    if world > 1:
        os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                          MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        sharding.init_distributed("gloo")
    t = GeneralizingTrainer2(_ToyModel, dict(_TOY_CFG), device="cpu")
    n = t.fit(_toy_data(), stages=stages, batch_size=2, rank=rank, world_size=world)
    flat = torch.cat([p.detach().reshape(-1) for p in t.model.parameters()])
    q.put((rank, n, flat.tolist(), list(t.model.calls), list(t.model.centres)))  # plain lists only
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _run_joint(world, stages):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_joint_worker, args=(r, world, port, q, stages)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted((q.get(timeout=100) for _ in range(world)), key=lambda r: r[0])
    for p in procs:
        p.join(timeout=30)
        assert p.exitcode == 0
    return [(r, n, torch.tensor(flat), calls, centres) for r, n, flat, calls, centres in res]


@pytest.mark.timeout(180)
def test_joint_trainer_single_process_follows_reference_loop():
    stages = [{"step1": 2, "step2": 1, "step3": 3}]
    (_, n, _, calls, _c), = _run_joint(1, stages)
    # 2 epochs x 2 batches of 2 images: step 1 on the batch, then per image step 2 and step 3
    per_batch = [(1, 2)] * 2 + ([(2, 1)] * 1 + [(3, 1)] * 3) * 2
    assert calls == per_batch * 4
    assert n == len(calls)


@pytest.mark.timeout(180)
def test_joint_trainer_data_parallel_matches_single_process():
    """Step 1 (batch-mean loss, whole-batch depth mean): two ranks with averaged gradients reproduce
    the single-process run; with steps 2 / 3 the ranks still hold identical parameters."""
    stages = [{"step1": 3, "step2": 0, "step3": 0}]
    (_, n1, ref, _, _c), = _run_joint(1, stages)
    res = _run_joint(2, stages)
    assert [r[1] for r in res] == [n1, n1]
    for _, _, flat, calls, _c in res:
        assert all(c == (1, 1) for c in calls)          # each rank sees one image of each batch
        assert float((flat - ref).abs().max()) < 1e-5
    res = _run_joint(2, [{"step1": 1, "step2": 2, "step3": 1}])
    assert torch.equal(res[0][2], res[1][2])


@pytest.mark.timeout(180)
def test_joint_trainer_depth_centre_scope_under_data_parallelism():
    """The whole-batch (all ranks) depth centre applies where the reference runs a BATCH of images
    through the depth net — step 1 (model.py:338) — and not to the per-image step 3, whose inner
    step-1 pass centres each image by its own mean (trainer.py:452, model.py:236) at any W."""
    res = _run_joint(2, [{"step1": 1, "step2": 1, "step3": 2}])
    for _, _, _, _, centres in res:
        s1 = [(c, own) for k, c, own in centres if k == 1]
        s3 = [(c, own) for k, c, own in centres if k == 3]
        assert s1 and s3
        assert all(c == own for c, own in s3)                 # per-image centre
        assert any(abs(c - own) > 1e-6 for c, own in s1)      # mean over both ranks' images
    # both ranks used the SAME centre in step 1
    assert [c for k, c, _ in res[0][4] if k == 1] == pytest.approx([c for k, c, _ in res[1][4] if k == 1])
