import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
import numpy as np, torch
import relnet_amd
from relnet_amd import ops, backbone, detector
from oracle import postprocess as OPP
H, W = 192, 256
p = backbone.init_params(seed=3)
g = torch.Generator().manual_seed(5)
for k in ('cls_score_weight', 'bbox_pred_weight'):
    p[k] = torch.randn(p[k].shape, generator=g) * 0.05
data = torch.randn(1, 3, H, W, generator=g)
im_info = torch.tensor([[H, W, 1.0]])
cfg = detector.Config(); cfg.rpn_post_nms_top_n = 100
det = detector.Detector(p, dtype=torch.float32, relation=False, im_hw=(H, W), cfg=cfg)
out = det.forward(data.cuda(), im_info.cuda())
prob = out['cls_prob'][0].float().cpu().numpy()
full = np.zeros((100, 8)); full[:, 4:8] = out['pred_boxes'][0].cpu().numpy()
want = OPP.detections(prob, full, 81, 1e-3, 0.6, True, 100)
raw = OPP.detections(prob, full, 81, 1e-3, 0.6, True, -1)
n = int(out['num_detections'][0])
got = out['detections'][0, :n].cpu().numpy()
flat = np.concatenate([np.hstack((np.full((len(w), 1), c + 1.0), w[:, 4:5], w[:, :4])) for c, w in enumerate(want)])
print('n', n, len(flat), 'thresh', float(out['image_thresh'][0]))
allsc = np.sort(np.hstack([r[:, 4] for r in raw]))
print('oracle thresh', allsc[-100], 'neighbors', allsc[-103:-97])
bad = np.where(np.abs(got - flat.astype(np.float32)).max(axis=1) > 1e-4)[0]
print('bad rows', bad[:10])
for i in bad[:6]:
    print(i, got[i], flat[i])
cnt = out['class_counts'][0].cpu().numpy()
print('counts eq', all(cnt[c] == len(raw[c]) for c in range(80)))
for c in range(80):
    d = out['class_dets'][0, c, :cnt[c]].cpu().numpy()
    if not np.allclose(d, raw[c], rtol=1e-9, atol=1e-12):
        k = np.where(np.abs(d - raw[c]).max(axis=1) > 1e-9)[0]
        print('class', c, 'first bad pick', k[:5], d[k[0]], raw[c][k[0]])
        break
u, ucnt = np.unique(prob[:, 1], return_counts=True)
print('dup probs in class1:', (ucnt > 1).sum(), 'unique rois', len(np.unique(out['rois'][0].cpu().numpy(), axis=0)))
