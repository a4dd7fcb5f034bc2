"""N>1 path on CPU: two gloo ranks run the sharding / barrier / max-over-ranks protocol that
bench.py uses on RCCL (images are independent units; there is no data-path collective)."""
import os
import socket

import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import relnet_amd  # noqa: F401
    from relnet_amd import dist as D
    r, w, _ = D.init(backend='gloo')
    lo, hi = D.shard_images(37, r, w)
    D.fence()
    elapsed = 0.5 + 0.25 * r                       # rank 1 is the slow one
    thr = D.throughput(hi - lo, elapsed)
    q.put((r, lo, hi, thr, D.max_over_ranks(elapsed)))
    D.fence()
    torch.distributed.destroy_process_group()


def test_two_rank_protocol():
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, lo0, hi0, thr0, mx0), (r1, lo1, hi1, thr1, mx1) = res
    assert (lo0, hi0, lo1, hi1) == (0, 19, 19, 37)             # disjoint, covering, balanced
    assert mx0 == mx1 == 0.75
    assert abs(thr0 - 37 / 0.75) < 1e-9 and thr0 == thr1       # whole-job rate / slowest rank


def test_shard_edges():
    import relnet_amd  # noqa: F401
    from relnet_amd.dist import shard_images
    assert [shard_images(5, r, 8) for r in range(8)] == [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5), (5, 5), (5, 5), (5, 5)]
    assert shard_images(16, 3, 4) == (12, 16)


def _grad_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import relnet_amd  # noqa: F401
    from relnet_amd import dist as D
    D.init(backend='gloo')
    shapes = [('conv1_weight', (64, 3, 7, 7)), ('res2a_branch2a_weight', (64, 64, 1, 1)), ('bn4b3_branch2a_gamma', (256,)),
              ('res4b3_branch2a_weight', (256, 1024, 1, 1)), ('fc_new_1_weight', (8, 50)), ('query_1_bias', (7,))]
    b = D.GradientBucket(shapes)
    for i, n in enumerate(b.names):
        b.view(n).fill_(float(rank + 1) * (i + 1))
    b.all_reduce()
    q.put((rank, b.names, [float(b.view(n).flatten()[0]) for n in b.names], int(b.flat.numel())))
    torch.distributed.destroy_process_group()


def test_gradient_bucket_sum_allreduce():
    """A13: frozen-by-substring rule + one flat SUM all-reduce (rescale_grad = 1 semantics)."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, names, vals, numel in res:
        assert names == ['res4b3_branch2a_weight', 'fc_new_1_weight', 'query_1_bias']   # conv1/res2/gamma frozen
        assert vals == [3.0, 6.0, 9.0]                                                 # (1 + 2) * (i + 1): SUM, not mean
        assert numel == 256 * 1024 + 448 + 64                                          # 64-element aligned slices


def _flat_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import relnet_amd  # noqa: F401
    from relnet_amd import dist as D
    from relnet_amd import train
    D.init(backend='gloo')
    g = torch.Generator().manual_seed(0)                        # same initial weights on every rank
    named = [('res4b3_branch2a', torch.randn(256, 1024, generator=g)), ('fc_new_2', torch.randn(8, 50, generator=g)),
             ('pair_pos_fc1_1', torch.randn(16, 64, generator=g))]
    flat = train._Flat(named, 'cpu')
    for i, (n, t) in enumerate(named):
        flat.view(flat.grad, n).fill_(float(rank + 1) * (i + 1))
    train.all_reduce_sum(flat.grad)
    q.put((rank, [float(flat.view(flat.grad, n).flatten()[0]) for n, _ in named], int(flat.size),
           bool(torch.equal(flat.view(flat.master, 'fc_new_2'), named[1][1])), [flat.slices[n][0] % 64 for n, _ in named]))
    torch.distributed.destroy_process_group()


def test_trainer_flat_buffers_sum_allreduce():
    """The training step's exchange (train.Trainer.all_reduce): flat fp32 gradient buffer, one SUM all-reduce, every rank
    ends with the sum; slices are 64-element aligned views of one buffer."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_flat_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, firsts, size, master_ok, align in res:
        assert firsts == [3.0, 6.0, 9.0]                        # (1 + 2) * (i + 1)
        assert size == 256 * 1024 + 448 + 1024 and master_ok and align == [0, 0, 0]
