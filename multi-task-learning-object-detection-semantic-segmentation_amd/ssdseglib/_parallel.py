"""Data parallelism over the 8 GPUs of a node: one process per GPU, replicated weights, ONE collective per step.

The reference has no distributed code; this is the new part.  Work shards by image: global batch
-> `world` contiguous shards, every rank runs the full step on its shard (BatchNorm statistics and the hard-negative
mining pool are per-replica, exactly what `tf.distribute.MirroredStrategy` would give the reference), then the flat fp32
gradient bucket (4,009,920 floats = 16 MB) is summed across ranks and Adam applies it scaled by 1/world, identically on
every rank -- so weights never need a broadcast after step 0.  The result equals the mean of `world` independent
reference steps taken from the same weights.

torch.distributed is plumbing only: backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU for tests.  xGMI is
point-to-point (7 links x ~153 GB/s), a 16 MB ring all-reduce is ~0.2 ms against a >= 25 ms step, so a single flat
bucket is the right granularity (no bucketing / overlap machinery is worth its launches here).
"""
from __future__ import annotations

import os
from typing import Optional, Tuple


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment (defaults: single process)."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def shard_bounds(global_batch: int, rank: int, world: int) -> Tuple[int, int]:
    """contiguous shard [begin, end) of `rank`; sizes differ by at most one when world does not divide the batch"""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(global_batch, world)
    begin = rank * base + min(rank, extra)
    return begin, begin + base + (1 if rank < extra else 0)


class GradientAllReduce:
    """Sums one flat gradient bucket (a torch tensor: CUDA for RCCL, CPU for gloo) across the process group.

    `__call__()` is enqueued on torch's current stream (the stream the HIP context borrows), so it is ordered after the
    backward kernels and before Adam without host synchronisation.  Adam then multiplies by `scale` = 1/world."""

    def __init__(self, bucket, group=None):
        import torch.distributed as dist
        self._dist = dist
        self.bucket = bucket
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.scale = 1.0 / self.world

    def __call__(self):
        if self.world > 1:
            self._dist.all_reduce(self.bucket, op=self._dist.ReduceOp.SUM, group=self.group)

    def check_replicas_in_sync(self, params) -> float:
        """max |p - p_rank0| over the group (debug aid: replicas must stay bit-identical)"""
        if self.world == 1:
            return 0.0
        ref = params.clone()
        self._dist.broadcast(ref, src=0, group=self.group)
        diff = (params - ref).abs().max()
        self._dist.all_reduce(diff, op=self._dist.ReduceOp.MAX, group=self.group)
        return float(diff)


def init_process_group(backend: Optional[str] = None, device_index: Optional[int] = None):
    """torch.distributed rendezvous from the torchrun environment; 'nccl' (= RCCL) when a device index is given."""
    import torch
    import torch.distributed as dist
    rank, local_rank, world = env_world()
    if world == 1:
        return None
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC only on this driver
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    backend = backend or ("nccl" if device_index is not None else "gloo")
    kwargs = {}
    if backend == "nccl":
        torch.cuda.set_device(device_index)
        kwargs["device_id"] = torch.device("cuda", device_index)
    dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return dist
