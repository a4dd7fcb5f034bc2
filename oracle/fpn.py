"""CPU restatement of the FPN configuration's extra pieces (TEST INFRASTRUCTURE ONLY).

  * roi_dispatch      relation_rcnn/core/rcnn.py:53-74 (cfg.network.ROIDispatch): pyramid level per roi and
                      the level-major regrouping, incl. the all-zero dummy roi of an empty level (:61-71)
  * fpn_neck          symbols/resnet_v1_101_rcnn_fpn_attention_1024_pairwise_position_multi_head_16_learn_nms.py
                      :804-840 (1x1 laterals, nearest 2x upsampling + sum, 3x3 output convs)
  * pool_levels       :1108-1121 (four ROIPooling calls at 1/4..1/32 + Concat(dim=0))

rcnn.py is plain numpy: the level formula here IS the reference's expression on float32 boxes, except that
log2 is evaluated correctly rounded (see oracle/relation.py:cr).  roi_dispatch is PINNED: tests/golden/fpn.npz holds the
output of the reference's own get_rcnn_testbatch on seeded float32 proposals (level boundaries, empty level with its
dummy roi), tests/test_oracle_golden.py::test_fpn_roi_dispatch_matches_reference_loader compares.  Convolution / UpSampling / ROIPooling are
MXNet built-ins: restated from v1.1.0 semantics, PARITY UNPINNED.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import roi_pooling as ORP
from .relation import cr

F32 = np.float32


def roi_levels(boxes):
    """boxes [N,4] float32 -> feat_id [N] int (rcnn.py:56-58)."""
    boxes = np.asarray(boxes, F32)
    w = boxes[:, 2] - boxes[:, 0] + F32(1)
    h = boxes[:, 3] - boxes[:, 1] + F32(1)
    s = (np.sqrt(w * h) / F32(224)).astype(F32)
    return np.clip(np.floor(F32(2) + cr(np.log2, s)), 0, 3).astype(int)


def roi_dispatch(boxes, dummy_for_empty=True):
    """-> (rois [N',5] level-major with batch index 0, level [N'], perm [N'] (-1 for a dummy row), counts [4])."""
    boxes = np.asarray(boxes, F32)
    lv = roi_levels(boxes)
    rois, levels, perm, counts = [], [], [], []
    for l in range(4):
        idx = np.where(lv == l)[0]
        counts.append(len(idx))
        r = boxes[idx]
        if len(idx) == 0 and dummy_for_empty:
            r = np.zeros((1, 4), F32); idx = np.array([-1])
        rois.append(np.hstack((np.zeros((len(r), 1), F32), r)).astype(F32))
        levels.append(np.full(len(r), l)); perm.append(idx)
    return np.vstack(rois), np.concatenate(levels), np.concatenate(perm), np.array(counts)


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32) if not torch.is_tensor(x) else x.float()


def fpn_neck(c2, c3, c4, c5, p, with_ft64=False):
    """res2c / res3b3 / res4b22 / res5c outputs -> (fpn_ft4, fpn_ft8, fpn_ft16, fpn_ft32[, fpn_ft64])."""
    conv = lambda x, n, **kw: F.conv2d(x, _t(p[n + '_weight']), _t(p[n + '_bias']), **kw)
    l32 = conv(c5, 'fpn_ft32_1x1')
    p16 = F.interpolate(l32, scale_factor=2, mode='nearest') + conv(c4, 'fpn_ft16_1x1')
    p8 = F.interpolate(p16, scale_factor=2, mode='nearest') + conv(c3, 'fpn_ft8_1x1')
    p4 = F.interpolate(p8, scale_factor=2, mode='nearest') + conv(c2, 'fpn_ft4_1x1')
    outs = (conv(p4, 'fpn_ft4_3x3', padding=1), conv(p8, 'fpn_ft8_3x3', padding=1),
            conv(p16, 'fpn_ft16_3x3', padding=1), conv(l32, 'fpn_ft32_3x3', padding=1))
    if with_ft64:
        outs = outs + (conv(l32, 'fpn_ft64_3x3', stride=2, padding=1),)
    return outs


def pool_levels(feats, rois, levels):
    """feats: 4 maps [B,C,H_l,W_l]; rois [R,5] level-major; levels [R] -> [R,C,7,7]."""
    out = []
    for l, sc in enumerate((1 / 4.0, 1 / 8.0, 1 / 16.0, 1 / 32.0)):
        sel = np.where(np.asarray(levels) == l)[0]
        if len(sel):
            out.append(ORP.roi_pooling(np.asarray(feats[l], F32), rois[sel], (7, 7), sc))
    return np.concatenate(out, 0)
