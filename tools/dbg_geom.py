"""debug: ln G of the learn-NMS geometry (FPN gradient test setup) -- libm float32 kernel vs matrix-core kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
import numpy as np, torch
import relnet_amd
from relnet_amd import backbone, train, ops
from relnet_amd.relation import pack_pair_pos
from test_gpu_fpn import _proposals
H, W, N, G = 128, 160, 60, 4
p = backbone.init_params(seed=41, fpn=True)
g = torch.Generator().manual_seed(42)
for k in ('cls_score_weight', 'bbox_pred_weight'):
    p[k] = torch.randn(p[k].shape, generator=g) * 0.05
for lvl in (4, 8, 16, 32):
    p['fpn_ft%d_1x1_weight' % lvl] = p['fpn_ft%d_1x1_weight' % lvl] * 2
    p['fpn_ft%d_3x3_weight' % lvl] = p['fpn_ft%d_3x3_weight' % lvl] * 2
    p['fpn_ft%d_3x3_bias' % lvl] = torch.rand(256, generator=g) * 0.1
p['nms_logit_bias'] = torch.zeros(5)
for k in ('nms_logit_weight', 'nms_rank_weight', 'roi_feat_embedding_weight', 'nms_query_1_weight', 'nms_key_1_weight',
          'nms_linear_out_1_weight', 'nms_pair_pos_fc1_1_weight'):
    p[k] = torch.randn(p[k].shape, generator=g) * 0.05
cfg = train.TrainConfig(); cfg.learn_nms, cfg.first_n = True, 24
data = torch.randn(1, 3, H, W, generator=g)
props = _proposals(N, 43, H, W)[None]
gt = np.zeros((1, G, 5), np.float32); gt[0, :, :4] = props[0, [8, 17, 29, 44]]; gt[0, :, 4] = [3, 17, 17, 60]
tr = train.FPNTrainer(p, cfg)
d = lambda a: torch.as_tensor(a).cuda()
out = tr.forward_backward(data.cuda(), torch.tensor([[H, W, 1.0]]).cuda(), d(gt), d(props))
cb = out['nms_class_boxes'][0]                     # [C, F, 4]
print('class_boxes', tuple(cb.shape), 'w min/max', float((cb[..., 2] - cb[..., 0] + 1).min()), float((cb[..., 2] - cb[..., 0] + 1).max()),
      'h min', float((cb[..., 3] - cb[..., 1] + 1).min()))
class M_: pass
mod = M_(); mod.wp = tr.W.view(tr.W.master, 'nms_pair_pos_fc1_1'); mod.bp = tr.b('nms_pair_pos_fc1_1')
wp_t, bp = pack_pair_pos([mod], 'cuda')
F = cb.shape[1]
b_libm = ops.geometry_bias(cb.contiguous(), wp_t, bp, F, fast32=True)[0][..., :F]
b_mfma = ops.geometry_bias(cb.contiguous(), wp_t, bp, F, mfma32=True)[0][..., :F]
b_exact = ops.geometry_bias(cb.contiguous(), wp_t, bp, F)[0][..., :F]
fl = float(np.log(1e-6))
for name, b in (('libm', b_libm), ('mfma', b_mfma)):
    dd = (b - b_exact).abs()
    act_e, act = b_exact > fl + 1e-3, b > fl + 1e-3
    print(name, 'max |d lnG|', float(dd.max()), 'mean', float(dd.mean()), 'clamp disagreements', int((act_e != act).sum()), 'of', act.numel(),
          'active frac', float(act_e.float().mean()))
    both = act_e & act
    print('   over both-active: max', float(dd[both].max()), 'p99', float(dd[both].flatten().kthvalue(int(0.99 * both.sum())).values))
    dG = (b.exp() - b_exact.exp()).abs()
    print('   max |dG|', float(dG.max()), 'mean |dG|', float(dG.mean()), 'mean G', float(b_exact.exp().mean()))
# worst pairs
dd = (b_mfma - b_exact).abs()
idx = torch.nonzero(dd == dd.max())[0].tolist()
c, h, i, j = idx
print('worst', idx, 'box i', cb[c, i].tolist(), 'box j', cb[c, j].tolist(), 'lnG exact/mfma', float(b_exact[c, h, i, j]), float(b_mfma[c, h, i, j]))
