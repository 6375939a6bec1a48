"""Data-parallel plumbing (new capability: the reference is single-device): frames of a sequence are
sharded over ranks in contiguous blocks, every rank keeps the full parameter arena, and the flat fp32 gradient bucket is
summed with ONE all-reduce per step (RCCL over xGMI on the GPU box: backend "nccl"; gloo in the CPU tests).  The 1/world
factor is applied by the Adam kernel (`grad_scale`), so mean-type losses average over ranks and frame-independent
regularisers (computed identically on every rank from the same RNG seed) are counted once."""
import torch
import torch.distributed as dist


def shard_frames(T, rank, world):
    """contiguous block [lo, hi) of the T frames owned by `rank` (T % world == 0 is required, like C4: 256 = 8 x 32)"""
    if T % world:
        raise ValueError(f"{T} frames do not split evenly over {world} ranks")
    per = T // world
    return rank * per, (rank + 1) * per


def batches(lo, hi, batch_size, step):
    """frame ids of this rank's `step`-th mini-batch: cycles through its block in order"""
    n = hi - lo
    return (torch.arange(batch_size) + step * batch_size) % n + lo


def allreduce_flat(bucket):
    """sum the flat gradient bucket over all ranks in one collective (no-op for a single process)"""
    if dist.is_available() and dist.is_initialized():
        dist.all_reduce(bucket)
    return bucket
