"""Data parallelism of the M1 train step: one process per GPU, batch sharded across ranks, ONE exchange per
step -- a sum all-reduce of the flat gradient buffer over RCCL/xGMI (``torch.distributed`` backend "nccl" on
ROCm), replacing the reference's single-process tf.distribute.MirroredStrategy (train_model.py:167-170).

Volumes are independent in forward and backward (InstanceNorm is per sample, the SE gate depends on
parameters only, Focal and KL are batch means), so the average of per-rank gradients equals the global-batch
gradient; the 1/world_size factor is folded into the optimiser kernel (``grad_scale``).

xGMI is point-to-point (7 links per GPU): the flat buffer is cut into a few large buckets that are issued
asynchronously back-to-back so RCCL can keep every link busy, instead of per-tensor collectives.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_process_group_from_env(backend: Optional[str] = None) -> int:
    """RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT from the environment (torchrun contract)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        dist.init_process_group(backend=backend, rank=int(os.environ.get("RANK", "0")), world_size=world)
    return world


def bucket_bounds(n: int, bucket_elems: int) -> List[tuple]:
    """[lo, hi) ranges covering [0, n); the last bucket absorbs the remainder."""
    if n <= 0:
        return []
    nb = max(1, n // max(1, bucket_elems))
    step = -(-n // nb)
    step = (step + 3) // 4 * 4
    out, lo = [], 0
    while lo < n:
        hi = min(n, lo + step)
        out.append((lo, hi))
        lo = hi
    return out


class GradReducer:
    """Bucketed asynchronous sum all-reduce of a flat gradient buffer."""

    def __init__(self, world_size: Optional[int] = None, bucket_mb: float = 64.0, group=None):
        self.group = group
        self.world_size = world_size if world_size is not None else (dist.get_world_size(group) if dist.is_initialized() else 1)
        self.bucket_elems = int(bucket_mb * (1 << 20) / 4)

    @property
    def grad_scale(self) -> float:
        return 1.0 / float(self.world_size)

    def all_reduce(self, flat_grad: torch.Tensor) -> None:
        if self.world_size <= 1 or not dist.is_initialized():
            return
        works = []
        for lo, hi in bucket_bounds(flat_grad.numel(), self.bucket_elems):
            works.append(dist.all_reduce(flat_grad[lo:hi], op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        for w in works:
            w.wait()


def shard_batch(global_batch: int, rank: int, world_size: int) -> range:
    """Indices of the global batch owned by ``rank`` (batch must divide evenly, train_model.py:170)."""
    assert global_batch % world_size == 0, \
        'Batch size (%d) should be a multiple of the number of GPUs (%d).' % (global_batch, world_size)
    per = global_batch // world_size
    return range(rank * per, (rank + 1) * per)
