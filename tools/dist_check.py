"""Two-rank check of the training step's gradient exchange on a ONE-GPU box (both ranks on cuda:0, gloo):
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 tools/dist_check.py
Both ranks get the SAME batch and weights, so after the bucketed all-reduce every gradient must equal world x the local one --
for the eager step (buckets announced from the backward pass) and for train.CapturedStep (buckets between graph segments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import __graft_entry__ as ge
ge.build()
import relnet_amd  # noqa: F401
from relnet_amd import backbone, train, dist as D

torch.cuda.set_device(0)
rank, world, _ = D.init(backend='gloo')
B, H, W, G = 2, 384, 512, 4
params = backbone.init_params(seed=1)
cfg = train.TrainConfig(); cfg.learn_nms = True
cfg.rank_in_anchor_seed = False        # identical batches AND identical anchor subsets on both ranks: the sum must be 2 x local
tr = train.Trainer(params, cfg, im_hw=(H, W))
g = torch.Generator().manual_seed(7)
data = torch.randn(B, 3, H, W, generator=g).cuda()
im_info = torch.tensor([[float(H), float(W), 1.0]] * B).cuda()
rng = np.random.default_rng(3)
gt = np.zeros((B, G, 5), np.float32)
for b in range(B):
    bw, bh = rng.uniform(32, 200, G), rng.uniform(32, 200, G)
    x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
    gt[b] = np.stack([x1, y1, x1 + bw, y1 + bh, rng.integers(1, 81, G)], 1)
batch = (data, im_info, torch.as_tensor(gt).cuda())
res = {}
def same_sample():            # the RPN target sampler draws from a per-step counter: every compared pass uses the same draw
    if getattr(tr, '_anchor_step', None) is not None:
        tr._anchor_step.zero_()


with torch.no_grad():
    tr.forward_backward(*batch); tr.all_reduce(); torch.cuda.synchronize()          # warm-up (creates the counter)
    same_sample()
    tr.forward_backward(*batch)
    torch.cuda.synchronize()
    # (forward_backward announced its buckets: finish the exchange, then compare with a second, un-exchanged pass)
    order = tr.all_reduce(); torch.cuda.synchronize()
    summed = tr.W.grad.clone(); bsum = tr.Bv.grad.clone()
    import torch.distributed as dist
    pg_world = dist.get_world_size()
    # un-exchanged pass: temporarily pretend there is one rank
    active = D.BucketedAllReduce.__dict__['active']          # the staticmethod object itself
    D.BucketedAllReduce.active = staticmethod(lambda: False)
    same_sample()
    tr.forward_backward(*batch); tr._grad_buckets().finish(); torch.cuda.synchronize()
    D.BucketedAllReduce.active = active
    local = tr.W.grad.clone(); blocal = tr.Bv.grad.clone()
    den = local.abs().max().item()
    res['eager_max_rel_diff'] = ((summed - pg_world * local).abs().max().item() / (pg_world * den))
    res['eager_bias_exact'] = bool(torch.allclose(bsum, pg_world * blocal, rtol=1e-5, atol=1e-6 * blocal.abs().max().item()))
    res['eager_order'] = order
    # captured step
    same_sample()
    step = train.CapturedStep(tr, batch)          # (the capture pass itself advances the counter once)
    same_sample()
    step.replay(); tr.all_reduce(); torch.cuda.synchronize()
    res['captured_max_rel_diff'] = ((tr.W.grad - pg_world * local).abs().max().item() / (pg_world * den))
    res['captured_segments'] = [i for _, i in step.segments]
    res['finite'] = bool(torch.isfinite(tr.W.grad).all())
    # the path bench.py times: replay (buckets announced between the graph segments), all_reduce(wait=False) launches whatever is
    # missing WITHOUT waiting, update() then waits bucket by bucket in launch order and runs SGD on each slice as soon as its own
    # sum has landed (the optimizer of the early buckets overlaps the collectives still in flight)
    same_sample()
    w_before = tr.W.master.clone()
    step.replay()
    res['overlap_launch_order'] = tr.all_reduce(wait=False)
    last = res['overlap_launch_order'][-1]
    res['last_bucket_done_when_update_starts'] = bool(tr._grad_buckets().is_completed(last))
    tr.update(); torch.cuda.synchronize()
    res['update_order'] = [i for i, _ in tr.update_order]
    res['weights_moved'] = bool((tr.W.master != w_before).any()) and bool(torch.isfinite(tr.W.master).all())
    # same batch, same weights, summed gradients: the two ranks must hold bit-identical weights after the step
    chk = tr.W.master.double().sum().reshape(1).clone()
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN); dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    res['weights_equal_across_ranks'] = bool((lo == hi).all())
if rank == 0:
    print('DIST_CHECK', res)
dist.barrier(); dist.destroy_process_group()
