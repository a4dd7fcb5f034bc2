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
        # a finite collective timeout: a rank that died (out of memory, an exception between two collectives) must turn the others'
        # wait into an error after minutes, not into a hang that loses the whole run (RELNET_DIST_TIMEOUT_S overrides)
        import datetime
        dist.init_process_group(backend, rank=rank, world_size=world,
                                timeout=datetime.timedelta(seconds=int(os.environ.get('RELNET_DIST_TIMEOUT_S', '600'))))
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


class BucketedAllReduce(object):
    """The gradient exchange of the training step: SUM all-reduce (MXNet rescale_grad = 1.0 semantics,
    train_end2end.py:167; the reference pushes / pulls one tensor at a time, core/module.py:569-591) of ONE flat
    gradient buffer in a few large contiguous buckets, each issued as soon as the backward pass has finished writing it.

    `bounds` are element offsets [b0 = 0 < b1 < ... < bn = numel]; bucket i = flat[b_i : b_{i+1}].  The trainer lays the
    buffer out in forward order (res3 | res4 | res5 | heads), so the backward pass completes the buckets from the last
    to the first: `ready(i)` launches bucket i asynchronously -- on CUDA from a side stream that waits for an event
    recorded on the compute stream, so RCCL's ring over xGMI (point-to-point links, ~153 GB/s each: the collective is
    per-link bound and wants few, large messages) overlaps the remaining backward kernels; `finish()` launches whatever
    was not announced and makes the caller's stream wait for all of them.  With no process group (or one rank) every
    call is a no-op.  Buckets are disjoint slices: a collective in flight never touches memory the backward pass still
    writes, whatever the order of `ready` calls (tests/test_dist_gloo.py proves order independence on 2 gloo ranks)."""

    def __init__(self, flat, bounds):
        assert flat.dim() == 1 and bounds[0] == 0 and bounds[-1] == flat.numel() and list(bounds) == sorted(set(bounds))
        self.flat, self.bounds = flat, list(bounds)
        self.n = len(self.bounds) - 1
        self.cuda = flat.is_cuda
        self.stream = torch.cuda.Stream(device=flat.device) if self.cuda else None
        self.pending, self.done, self.launch_order = {}, set(), []
        self.exchanged = False      # True from finish() / clear() until the next reset(): this pass's sums are in the flat buffer

    @staticmethod
    def active():
        return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1

    def bucket(self, i):
        return self.flat[self.bounds[i]:self.bounds[i + 1]]

    def ready(self, i):
        """Bucket i holds its final local gradients (everything queued on the current stream so far)."""
        if i in self.done:
            return
        if self.cuda and self.active() and torch.cuda.is_current_stream_capturing():
            return          # inside a hipGraph capture the collective stays out of the graph: finish() issues it after the replay
        self.done.add(i)
        self.launch_order.append(i)
        if not self.active():
            return
        if self.cuda:
            ev = torch.cuda.Event()
            ev.record()                                     # compute stream: all writers of bucket i are queued before this
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                work = dist.all_reduce(self.bucket(i), op=dist.ReduceOp.SUM, async_op=True)
        else:
            work = dist.all_reduce(self.bucket(i), op=dist.ReduceOp.SUM, async_op=True)
        self.pending[i] = work

    def reset(self):
        """Start of a new backward pass: nothing announced yet (collectives of an abandoned pass are waited for first)."""
        for w in self.pending.values():
            w.wait()
        self.pending, self.done, self.launch_order = {}, set(), []
        self.exchanged = False

    def launch_rest(self):
        """Launch the buckets nobody announced (in backward order) WITHOUT waiting; returns the launch order so far."""
        for i in reversed(range(self.n)):
            self.ready(i)
        return list(self.launch_order)

    def pending_order(self):
        """Buckets whose collective was launched and not yet waited for, in launch order."""
        return [i for i in self.launch_order if i in self.pending]

    def is_completed(self, i):
        w = self.pending.get(i)
        return True if w is None else bool(w.is_completed())

    def wait(self, i):
        """The caller's stream waits for bucket i's collective only (RCCL: a stream dependency, the host does not block)."""
        w = self.pending.pop(i, None)
        if w is None:
            return
        w.wait()
        if self.cuda and dist.get_backend() != 'nccl':
            torch.cuda.synchronize()                        # gloo on device tensors (one-GPU test path), see finish()

    def clear(self):
        """End of a step whose buckets were waited for one by one."""
        assert not self.pending, sorted(self.pending)
        self.done, self.launch_order = set(), []
        self.exchanged = True

    def finish(self):
        """Launch the buckets nobody announced (in backward order) and wait: after this the flat buffer holds the sums."""
        for i in reversed(range(self.n)):
            self.ready(i)
        had = bool(self.pending)
        for w in self.pending.values():
            w.wait()                                        # CUDA: the current stream waits for the collective's stream
        if had and self.cuda and dist.get_backend() != 'nccl':
            # gloo on device tensors (the one-GPU test path): its copy-back runs on private streams whose events may not be
            # recorded yet when wait() returns -- order the device explicitly (RCCL's wait() is a proper stream dependency)
            torch.cuda.synchronize()
        order = self.launch_order
        self.pending, self.done, self.launch_order = {}, set(), []
        self.exchanged = True
        return order


def all_reduce_async(buf):
    """SUM all-reduce of `buf` launched asynchronously (None without a process group)."""
    if not BucketedAllReduce.active():
        return None
    return dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=True)


def wait_work(work, buf=None):
    if work is None:
        return
    work.wait()
    if buf is not None and buf.is_cuda and dist.get_backend() != 'nccl':
        torch.cuda.synchronize()
