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


# ---------------------------------------------------------------------------------------
# training: gradient exchange (reference: MXNet KVStore('device') push/pull per tensor,
# core/module.py:569-591, rescale_grad = 1.0 -> gradients are SUMMED over devices)
# ---------------------------------------------------------------------------------------
FIXED_PARAMS = ('conv1', 'bn_conv1', 'res2', 'bn2', 'gamma', 'beta')       # cfgs/*.yaml:23-29


def is_trainable(name, fixed=FIXED_PARAMS):
    """Frozen when ANY fixed pattern is a substring of the name (core/module.py:753-764)."""
    return not any(f in name for f in fixed)


class GradientBucket(object):
    """All trainable gradients in ONE flat buffer, reduced with a single all-reduce(SUM) per step
    instead of one push/pull per tensor: on xGMI (point-to-point links, ~153 GB/s each) a ring
    all-reduce is per-link bound, so the 68.3 M-element payload (273 MB fp32 / 137 MB bf16) wants one
    large collective, issued as soon as the backward pass has filled the buffer."""

    def __init__(self, named_shapes, dtype=torch.float32, device='cpu', fixed=FIXED_PARAMS):
        self.names = [n for n, _ in named_shapes if is_trainable(n, fixed)]
        self.shapes = {n: tuple(s) for n, s in named_shapes}
        self.offsets, off = {}, 0
        for n in self.names:
            self.offsets[n] = off
            numel = 1
            for d in self.shapes[n]:
                numel *= d
            off += (numel + 63) // 64 * 64                 # 64-element alignment of every slice
        self.flat = torch.zeros(off, dtype=dtype, device=device)

    def view(self, name):
        numel = 1
        for d in self.shapes[name]:
            numel *= d
        o = self.offsets[name]
        return self.flat[o:o + numel].view(self.shapes[name])

    def all_reduce(self, async_op=False):
        """SUM over ranks (MXNet rescale_grad = 1.0 semantics, train_end2end.py:167)."""
        if not dist.is_initialized():
            return None
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, async_op=async_op)
