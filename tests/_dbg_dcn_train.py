import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import test_gpu_train_step as TS
from oracle import train_graph as OT
dcn = int(sys.argv[1]) if len(sys.argv) > 1 else 1
H, W, G = 128, 160, 4
p, cfg, data, gt, L, Tg, Wg, train = TS._setup(H, W, G, 31, bool(dcn))
cfg.learn_nms, cfg.first_n, cfg.dcn = False, 24, bool(dcn)
tr = train.Trainer(p, cfg, im_hw=(H, W))
d = lambda a: torch.as_tensor(a).cuda()
out = tr.forward_backward(data.cuda(), torch.tensor([[H, W, 1.0]]).cuda(), d(gt), d(L[None]), d(Tg[None]), d(Wg[None]))
rois = out['rois'][0].cpu().numpy()
N = cfg.rpn_post_nms_top_n
pt = {k: v.double().clone() for k, v in p.items()}
with torch.no_grad():
    loss, parts = OT.total_loss(data.numpy(), pt, rois, out['label'][0].cpu().numpy(), out['bbox_target'][0].cpu().numpy(),
                                out['bbox_weight'][0].cpu().numpy(), L, Tg, Wg, N, dcn=bool(dcn))
dbg = out['debug']
def rel(a, b):
    a = a.double().cpu(); return float((a - b).abs().max() / b.abs().max()), float(((a - b).norm() / b.norm()))
print('conv5', rel(dbg['conv5'][0].permute(2, 0, 1), parts['conv5'][0]))
print('feat', rel(dbg['feat'][0].permute(2, 0, 1), parts['feat'][0]))
R = rois.shape[0]
po = parts['pooled'].permute(0, 2, 3, 1).reshape(R, -1)
print('pooled', rel(dbg['pooled'], po))
if dcn:
    print('trans', rel(dbg['trans'], parts['trans']), float(parts['trans'].abs().max()))
print('f1', rel(dbg['f1'][0], parts['f1']))
print('x2', rel(dbg['x2'][0], parts['x2']))
print('cls', rel(out['cls_score'][0], parts['cls_score']))
