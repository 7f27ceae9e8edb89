"""Episode-sharded data parallelism.

Episodes are independent units: every per-episode quantity depends only on the
episode and the shared parameters (model/common.py:368, trainer/utils.py:272-277
of the reference), so ranks own contiguous blocks of the episode stream and the
data path needs no collective.  The only exchanges are

  * training: ONE all-reduce (average) of a flat fp32 gradient bucket per step;
    loss_k is a mean over the local batch (trainer/utils.py:277), so with equal
    per-rank batches the averaged gradient equals the gradient of the global batch;
  * evaluation: ONE integer all-reduce of the metric counts at the end of the run
    (denet_amd.metrics.FullIOU.all_reduce / BinaryIOU.all_reduce).

One process per GPU; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU.
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Join the process group torchrun describes (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_*).
    Returns (rank, world, local_rank).  World size 1 needs no group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def world():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def shard_range(total: int, rank_: int, world_: int):
    """Contiguous block [lo, hi) of `total` episodes owned by `rank_` (sizes differ by at most one,
    earlier ranks take the extra)."""
    base, extra = divmod(total, world_)
    lo = rank_ * base + min(rank_, extra)
    return lo, lo + base + (1 if rank_ < extra else 0)


def shard_batch(batch, rank_: int = None, world_: int = None):
    """Slice every tensor of an episode batch (tuple / list / dict, leading dim = episodes)."""
    rank_ = rank() if rank_ is None else rank_
    world_ = world() if world_ is None else world_
    if world_ == 1:
        return batch

    def cut(t):
        lo, hi = shard_range(t.shape[0], rank_, world_)
        return t[lo:hi]

    if isinstance(batch, dict):
        return {k: cut(v) for k, v in batch.items()}
    return type(batch)(cut(t) for t in batch)


class GradBucket:
    """Flat fp32 view over the gradients of `params`: one collective per step instead of one per
    tensor (7 estimator tensors = 0.81-0.87 MB; ~21 MB with the conv stack: SURVEY.md §5)."""

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        n = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        off = 0
        for p in self.params:
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def all_reduce_mean(self, group=None):
        """Average over ranks (sum then divide: AVG is not available on gloo)."""
        w = world()
        if w > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=group)
            self.flat.div_(w)


def sync_buffers(module, group=None):
    """Make the module's buffers identical on every rank: floating-point ones (BatchNorm running_mean /
    running_var, which every rank updated from its own shard's statistics) become the mean over ranks,
    integer ones (num_batches_tracked) are taken from rank 0.  Called before validation and checkpoints, so
    that every rank evaluates -- and rank 0 saves -- ONE model instead of rank 0's shard-local statistics.
    One flat all-reduce for the float buffers."""
    if world() == 1:
        return
    fl = [b for b in module.buffers() if b.is_floating_point()]
    if fl:
        flat = torch.cat([b.detach().reshape(-1).float() for b in fl])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world())
        off = 0
        for b in fl:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()
    for b in module.buffers():
        if not b.is_floating_point():
            dist.broadcast(b, src=0, group=group)
