"""Multi-GPU plumbing: one process per GPU, images sharded across ranks, NO data-path collective.

The reference's only parallelism is data parallel with one image per device
(relation_rcnn/core/DataParallelExecutorGroup.py:336-361; inference results are gathered on the
host, core/tester.py:40).  Inference here is the same: ranks are replicas working on disjoint
images.  The only communication is the measurement protocol of bench.py (barrier, max of the
elapsed time) and, for evaluation drivers, a gather of per-rank detection counts.
backend 'nccl' is RCCL on ROCm; 'gloo' is used by the CPU tests.
"""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise torch.distributed from the torchrun environment (RANK/WORLD_SIZE/MASTER_*);
    returns (rank, world, local_rank).  World size 1 needs no process group."""
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        backend = backend or ('nccl' if torch.cuda.is_available() else 'gloo')
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def shard_images(num_images, rank, world):
    """Contiguous shard [lo, hi) of image indices for this rank (independent units)."""
    per, extra = divmod(num_images, world)
    lo = rank * per + min(rank, extra)
    return lo, lo + per + (1 if rank < extra else 0)


def fence(device=None):
    """Barrier bracketed by device synchronisation (both sides of a timed region)."""
    if device is not None and torch.cuda.is_available():
        torch.cuda.synchronize()
    if dist.is_initialized():
        dist.barrier()
        if device is not None and torch.cuda.is_available():
            torch.cuda.synchronize()


def max_over_ranks(value, device='cpu'):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device='cpu'):
    if not dist.is_initialized():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def throughput(images_this_rank, elapsed_this_rank, device='cpu'):
    """Whole-job images/s: all images of all ranks / the slowest rank's time."""
    return sum_over_ranks(images_this_rank, device) / max_over_ranks(elapsed_this_rank, device)
