"""Multi-step two-rank diagnosis (both ranks on cuda:0, gloo): different data per rank, captured training step + update, and after
every exchange the ranks compare checksums of their all-reduced gradients (must be identical) and print their magnitude."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
import __graft_entry__ as ge
ge.build()
import relnet_amd  # noqa: F401
from relnet_amd import backbone, train, dist as D

torch.cuda.set_device(0)
rank, world, _ = D.init(backend='gloo')
mode = os.environ.get('MODE', 'captured')
B, H, W, G = 8, 600, 1000, 8
params = backbone.init_params(seed=1)
cfg = train.TrainConfig(); cfg.learn_nms = True
cfg.lr = float(os.environ.get('LR', '0.00025'))
tr = train.Trainer(params, cfg, im_hw=(H, W))
g = torch.Generator().manual_seed(1000 + rank)
data = torch.randn(B, 3, H, W, generator=g).cuda()
im_info = torch.tensor([[float(H), float(W), 1.0]] * B).cuda()
rng = np.random.default_rng(2 + rank)
gt = np.zeros((B, G, 5), np.float32)
for b in range(B):
    bw, bh = rng.uniform(32, 400, G), rng.uniform(32, 400, G)
    x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
    gt[b] = np.stack([x1, y1, x1 + bw, y1 + bh, rng.integers(1, 81, G)], 1)
batch = (data, im_info, torch.as_tensor(gt).cuda())


def report(tag):
    torch.cuda.synchronize()
    gsum = tr.W.grad.double().sum().item(); gabs = tr.W.grad.abs().max().item()
    fin = bool(torch.isfinite(tr.W.grad).all())
    t = torch.tensor([gsum, gabs, float(fin), tr.W.master.abs().max().item()], dtype=torch.float64)
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    if rank == 0:
        same = all(torch.equal(o[:3], out[0][:3]) for o in out)
        print('STEP %-12s grads identical on all ranks: %s  max|g| %.3e  finite %s  max|w| %.3e' % (tag, same, out[0][1].item(), bool(out[0][2].item()), out[0][3].item()), flush=True)


with torch.no_grad():
    for i in range(2):
        tr.forward_backward(*batch); tr.all_reduce(); report('eager %d' % i); tr.update()
    step = train.CapturedStep(tr, batch) if mode == 'captured' else None
    for i in range(8):
        if step is not None:
            step.replay()
        else:
            tr.forward_backward(*batch)
        tr.all_reduce(); report('%s %d' % (mode, i)); tr.update()
dist.barrier(); dist.destroy_process_group()
