"""Data parallelism for the SSD training / inference step: one process per GPU, `torch.distributed`
(backend "nccl" == RCCL over xGMI on ROCm; "gloo" on CPU for tests).

The reference has no multi-device code at all.  What shards here:
  * inference (forward + decode) and target encoding are per-image independent -> every rank processes its own
    slice of the batch, no data-path collective ("replicas");
  * the training step needs exactly one collective, the gradient all-reduce (26.3 M fp32 parameters = 105 MB
    for SSD300/VOC).  xGMI is point-to-point (7 links x ~153 GB/s per GPU), so a ring is per-link bound:
    buckets of ~25 MB keep 4-5 collectives in flight behind the backward pass instead of one 105 MB tail.
  * hard-negative mining and the 1 / n_positives normalisation stay local to the rank's batch -- exactly the reference
    evaluated on that shard; DDP then AVERAGES the per-rank gradients, so a step is the mean over ranks of the reference's
    per-shard losses, not the reference's loss on the concatenated batch (that would need an all-reduce of n_positives and a
    distributed k-th-value select).  `tests/test_ddp_loss_gpu.py` pins this identity on the real kernels (two ranks, HIP
    encoder + HIP SSDLoss per shard, all-reduced gradient == mean of the oracle's per-shard gradients); `tests/test_distributed_cpu.py`
    pins the all-reduce arithmetic itself on CPU with a stand-in loss (SSDLoss has no CPU path).
"""
from __future__ import annotations

import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialise the default process group from RANK / WORLD_SIZE / MASTER_* (rendezvous on 127.0.0.1 unless told
    otherwise: container hostnames may not resolve).  Returns (rank, world_size, local_rank)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on these hosts
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local_rank


def shard_range(n_items, rank, world):
    """Contiguous slice [lo, hi) of a global batch owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def data_parallel(model, device=None, bucket_cap_mb=25):
    """Wrap `model` so backward all-reduces (averages) gradients in ~bucket_cap_mb buckets overlapped with the
    remaining backward computation.  No-op for a single process."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return model
    from torch.nn.parallel import DistributedDataParallel as DDP
    ids = [device.index] if (device is not None and device.type == "cuda") else None
    # A model with buffers (SSD7's BatchNorm running statistics) keeps them identical on every rank: rank 0's are broadcast at
    # each forward, so a checkpoint written by any rank describes the same model.  SSD300 / SSD512 have no buffers: no traffic.
    has_buffers = any(True for _ in model.buffers())
    return DDP(model, device_ids=ids, bucket_cap_mb=bucket_cap_mb, gradient_as_bucket_view=True,
               broadcast_buffers=has_buffers)


def max_over_ranks(value, device="cpu"):
    """Max of a python float over all ranks (the bench's step time)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device="cpu"):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
