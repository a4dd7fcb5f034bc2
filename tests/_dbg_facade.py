import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import relnet_amd
from relnet_amd import mx, backbone, detector
SYM_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'symbols')
H, W = 600, 1000
p = backbone.init_params(seed=1)
g = torch.Generator().manual_seed(101)
data = torch.randn(1, 3, H, W, generator=g); im_info = torch.tensor([[H, W, 1.0]])
sym = mx.sym.load(os.path.join(SYM_DIR, 'rcnn_end2end_relation_8epoch_test.json'))
it = sym.get_internals()
names = ['pool1_output', 'res2a_relu_output', 'res2c_relu_output', 'res3b3_relu_output', 'res4b22_relu_output', 'res5c_relu_output', 'conv_new_1_relu_output',
         'rpn_cls_score_output', 'rpn_bbox_pred_output', 'rpn_cls_prob_reshape_output', 'rois_output']
avail = it.list_outputs()
names = [n for n in names if n in avail]
grp = mx.sym.Group([it[n] for n in names])
exe = grp.bind(mx.gpu(0), args={k: p[k] for k in grp.list_arguments() if k in p}, aux_states={k: p[k] for k in grp.list_auxiliary_states()}, dtype=torch.bfloat16)
print(exe.fused_report)
outs = dict(zip(names, [o.data for o in exe.forward(is_train=False, data=data, im_info=im_info)]))
det = detector.Detector(p, dtype=torch.bfloat16, im_hw=(H, W))
f = det.backbone.forward(data.cuda())
ref = det.forward(data.cuda(), im_info.cuda())
def cmp(a, b, n):
    a, b = a.float(), b.float()
    print('%-32s shape %s rel_l2 %.3e max %.3e (ref max %.3e)' % (n, tuple(a.shape), ((a-b).norm()/b.norm()).item(), (a-b).abs().max().item(), b.abs().max().item()))
cmp(outs['res4b22_relu_output'], f['conv4'], 'conv4'); cmp(outs['res5c_relu_output'], f['conv5'], 'conv5')
cmp(outs['conv_new_1_relu_output'], f['conv_new_1_relu'], 'conv_new_1'); cmp(outs['rpn_cls_score_output'], f['rpn_cls_score'], 'rpn_cls'); cmp(outs['rpn_bbox_pred_output'], f['rpn_bbox_pred'], 'rpn_box')
r, rr = outs['rois_output'].cpu().numpy(), ref['rois'][0].cpu().numpy()
print(r[:5]); print(rr[:5])
print('same rows', (np.abs(r-rr).max(1)==0).sum())
