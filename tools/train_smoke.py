"""Small training-step smoke run (GPU): python tools/train_smoke.py [B] [H] [W]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import relnet_amd
from relnet_amd import backbone, train
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H = int(sys.argv[2]) if len(sys.argv) > 2 else 192
W = int(sys.argv[3]) if len(sys.argv) > 3 else 256
p = backbone.init_params(seed=3)
cfg = train.TrainConfig()
if H < 600:
    cfg.rpn_post_nms_top_n = 64
tr = train.Trainer(p, cfg, im_hw=(H, W))
g = torch.Generator().manual_seed(0)
data = torch.randn(B, 3, H, W, generator=g).cuda()
im_info = torch.tensor([[H, W, 1.0]] * B).cuda()
rng = np.random.default_rng(1)
G = 5
gt = np.zeros((B, G, 5), np.float32)
for b in range(B):
    x1 = rng.uniform(0, W - 80, G); y1 = rng.uniform(0, H - 80, G)
    gt[b, :, 0] = x1; gt[b, :, 1] = y1; gt[b, :, 2] = x1 + rng.uniform(30, 79, G); gt[b, :, 3] = y1 + rng.uniform(30, 79, G)
    gt[b, :, 4] = rng.integers(1, 81, G)
fh, fw = (H + 15) // 16, (W + 15) // 16
fh = -(-(-(-H // 2) - 1) // 2) ; fw = 0
# feature size of conv4: stem /2 (ceil), pool /2 (ceil mode), res3 /2, res4 /2
def fs(n):
    n = (n + 2 * 3 - 7) // 2 + 1
    n = -(-(n - 3) // 2) + 1
    n = (n - 1) // 2 + 1
    n = (n - 1) // 2 + 1
    return n
fh, fw = fs(H), fs(W)
labs, tgts, wgts = [], [], []
for b in range(B):
    L, T_, W_ = train.assign_anchor((fh, fw), gt[b], (H, W), cfg, seed=b)
    labs.append(L); tgts.append(T_); wgts.append(W_)
rl = torch.as_tensor(np.stack(labs)).cuda(); rt = torch.as_tensor(np.stack(tgts)).cuda(); rw = torch.as_tensor(np.stack(wgts)).cuda()
print('feat', fh, fw, 'rpn fg', int((rl == 1).sum()), 'bg', int((rl == 0).sum()))
gtd = torch.as_tensor(gt).cuda()
for it in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    out = tr.step(data, im_info, gtd, rl, rt, rw)
    torch.cuda.synchronize()
    gn = float(tr.W.grad.norm()); fin = bool(torch.isfinite(tr.W.grad).all() and torch.isfinite(tr.W.master).all())
    print('it', it, 'ms %.1f' % ((time.time() - t0) * 1e3), 'bbox_loss %.4f rpn_bbox_loss %.4f ohem %d' % (float(out['bbox_loss']), float(out['rpn_bbox_loss']), int(out['num_ohem'])), 'grad norm %.4g finite %s' % (gn, fin))
print('trainable', tr.num_trainable())
