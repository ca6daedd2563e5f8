"""Multi-GPU helpers: one process per GPU, torch.distributed (backend "nccl" == RCCL over xGMI on
ROCm; "gloo" on CPU for tests).

The reference has no distributed code on the GAN2Shape path.  Two modes are added:

  instance mode (default)   each image is optimised independently (trainer.py:75-127), so images
      are dealt round-robin to ranks and NO data-path collective is needed; only the job
      throughput is reduced at the end (max time, sum of units).
  joint mode                data-parallel training over the image batch (GeneralizingTrainer2,
      trainer.py:338-479): ONE bucketed gradient all-reduce (mean) per optimiser step — step 1:
      A (17 MB), step 2: E (55 MB), step 3: L+V+D+A (90 MB) — plus a scalar all-reduce so that
      get_clamped_depth centres depth by the whole-batch mean (model.py:338) rather than a
      per-rank mean.  xGMI is point-to-point (7 links per GPU): a single flat bucket per optimiser
      lets RCCL pick its direct / tree algorithms over all links instead of many small rings.
"""
import os

import torch
import torch.distributed as dist


def shard_indices(n_items, rank, world_size):
    """Images of rank `rank`: rank, rank + W, rank + 2W, ..."""
    return list(range(rank, n_items, world_size))


def init_distributed(backend=None):
    """Reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* from the environment (torch.distributed.run).
    Returns (rank, world_size, local_rank); no-op for a single process."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def job_throughput(units_local, seconds_local, device="cpu"):
    """Whole-job rate: units of all ranks / the slowest rank's time."""
    t = torch.tensor([float(seconds_local)], dtype=torch.float64, device=device)
    u = torch.tensor([float(units_local)], dtype=torch.float64, device=device)
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()) / float(t.item()), float(t.item())


def allreduce_mean_gradients(params):
    """Joint mode: average the gradients of `params` over all ranks with ONE collective (flat
    bucket), in place.  Parameters without a gradient contribute zeros."""
    if not (dist.is_initialized() and dist.get_world_size() > 1):
        return
    params = [p for p in params if p.requires_grad]
    if not params:
        return
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(dist.get_world_size())
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        if p.grad is None:
            p.grad = flat[off:off + n].view_as(p).clone()
        else:
            p.grad.copy_(flat[off:off + n].view_as(p))
        off += n


def global_mean(x):
    """Mean of `x` over its elements AND over all ranks (equal element counts per rank): the
    whole-batch mean of get_clamped_depth (model.py:338) under data parallelism."""
    m = x.mean()
    if dist.is_initialized() and dist.get_world_size() > 1:
        # differentiable all-reduce (its backward all-reduces the incoming gradient): every rank's
        # loss depends on every rank's depth through the shared centre, and after the gradient
        # averaging of allreduce_mean_gradients the result is the gradient of the batch-mean loss
        import torch.distributed.nn.functional as dist_fn
        m = dist_fn.all_reduce(m.clone(), op=dist.ReduceOp.SUM) / dist.get_world_size()
    return m
