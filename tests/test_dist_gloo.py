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


def _bucket_worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import relnet_amd  # noqa: F401
    from relnet_amd import dist as D
    D.init(backend='gloo')
    n = 4096
    bounds = [0, 1024, 1600, 3008, n]                              # four uneven buckets ("res3 | res4 | res5 | heads")
    results = []
    for order in ((3, 2, 1, 0), (3, 2), (0, 1, 2, 3), ()):         # backward order, partial, wrong-way-round, none announced
        g = torch.Generator().manual_seed(100 + rank)
        local = torch.randn(n, generator=g)
        flat = torch.zeros(n)
        bk = D.BucketedAllReduce(flat, bounds)
        bk.reset()
        if order == (0, 1, 2, 3):                                   # everything written, announced first to last
            flat.copy_(local)
            for i in order:
                bk.ready(i)
        else:
            for i in reversed(range(4)):                           # the "backward pass" fills the buckets last to first ...
                flat[bounds[i]:bounds[i + 1]] = local[bounds[i]:bounds[i + 1]]
                if i in order:                                      # ... and announces (some of) them right away, while the
                    bk.ready(i)                                     # lower buckets are still being written
        launched = bk.finish()
        want = sum(torch.randn(n, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
        results.append((list(order), launched, bool(torch.equal(flat, want))))
    # equals ONE all-reduce of the whole buffer (the round-1 exchange)
    whole = torch.randn(n, generator=torch.Generator().manual_seed(100 + rank))
    torch.distributed.all_reduce(whole)
    results.append(('whole', [], bool(torch.equal(whole, want))))
    q.put((rank, results))
    torch.distributed.destroy_process_group()


def test_bucketed_allreduce_order_and_overlap():
    """Training exchange (dist.BucketedAllReduce, used by train.Trainer): buckets announced while the rest of the buffer
    is still being written give exactly the SUM over ranks (rescale_grad = 1 semantics), for the backward order, a
    partial announcement (finish() launches the rest, last to first), the reverse order and no announcement at all."""
    world, port = 2, _free_port()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    procs = [ctx.Process(target=_bucket_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, results in res:
        for order, launched, ok in results:
            assert ok, (rank, order)
        assert results[0][1] == [3, 2, 1, 0] and results[2][1] == [0, 1, 2, 3]     # launch order = announcement order
        assert results[1][1][:2] == [3, 2] and sorted(results[1][1]) == [0, 1, 2, 3]      # announced first, the rest by finish()
        assert results[3][1] == [3, 2, 1, 0]                                               # finish() alone: backward order


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
