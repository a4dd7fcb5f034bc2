"""Oracle: the conv backbone / RPN head (torch-CPU float32) and the composition of the
whole test graph.  TEST INFRASTRUCTURE ONLY (also the `cpu_baseline` leg of bench.py).

Follows relation_rcnn/symbols/resnet_v1_101_rcnn_base.py:29-619 (conv1..conv4), :621-683
(conv5, dilate 2), :685-693 (RPN head) and the test branch of
resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16.py:176-322 /
resnet_v1_101_rcnn.py:96-174.  Convolution / BatchNorm / Pooling are MXNet built-ins
(un-vendored): restated from their v1.1.0 semantics -- PARITY UNPINNED.  BatchNorm is
applied as its own op (not folded) exactly as the graph spells it.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import proposal as OP
from . import roi_pooling as ORP
from . import relation as OR
from . import postprocess as OPP
from . import deform as OD

EPS = 1e-5
UNITS = (3, 4, 23, 3)
FILTERS = (256, 512, 1024, 2048)


def _t(x):
    return torch.as_tensor(np.asarray(x), dtype=torch.float32) if not torch.is_tensor(x) else x.float()


def _conv_bn(x, p, conv, bn, stride=1, pad=0, dil=1, relu=False):
    y = F.conv2d(x, _t(p[conv + '_weight']), None, stride=stride, padding=pad, dilation=dil)
    g, b = _t(p[bn + '_gamma']), _t(p[bn + '_beta'])
    m, v = _t(p[bn + '_moving_mean']), _t(p[bn + '_moving_var'])
    y = (y - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)
    return F.relu(y) if relu else y


def _unit_name(stage, u, n):
    if stage in (2, 5) or n <= 3:
        return '%d%s' % (stage, 'abc'[u])
    return '%d%s' % (stage, 'a' if u == 0 else 'b%d' % u)


def _bn(y, p, bn):
    g, b = _t(p[bn + '_gamma']), _t(p[bn + '_beta'])
    m, v = _t(p[bn + '_moving_mean']), _t(p[bn + '_moving_var'])
    return (y - m.view(1, -1, 1, 1)) / torch.sqrt(v.view(1, -1, 1, 1) + EPS) * g.view(1, -1, 1, 1) + b.view(1, -1, 1, 1)


def deformable_2b(y, p, nm):
    """res5x_branch2b of the DCN graphs (symbols/resnet_v1_101_rcnn_dcn_..._learn_nms.py:700-707): 72-channel
    offset conv (3x3, pad 2, dilate 2, bias) -> DeformableConvolution(num_deformable_group=4, no_bias) -> BN -> ReLU."""
    name = 'res%s_branch2b' % nm
    off = F.conv2d(y, _t(p[name + '_offset_weight']), _t(p[name + '_offset_bias']), padding=2, dilation=2)
    z = OD.deformable_convolution(y.numpy(), off.numpy(), _t(p[name + '_weight']).numpy(), None, (3, 3), (1, 1), (2, 2), (2, 2), 4)
    return F.relu(_bn(torch.as_tensor(z), p, 'bn%s_branch2b' % nm))


def res5(conv4, p, dcn=False, fpn=False):
    """conv4 [B,1024,h,w] -> conv5 [B,2048,h,w] (get_resnet_v1_conv5, dilate 2, stride 1); fpn: stride 2 on
    the first 1x1s and no dilation (symbols/resnet_v1_101_rcnn_fpn_..._learn_nms.py:707-720)."""
    x = conv4
    for u in range(3):
        nm = '5' + 'abc'[u]
        st = 2 if (fpn and u == 0) else 1
        dil = 1 if fpn else 2
        sc = _conv_bn(x, p, 'res%s_branch1' % nm, 'bn%s_branch1' % nm, stride=st) if u == 0 else x
        y = _conv_bn(x, p, 'res%s_branch2a' % nm, 'bn%s_branch2a' % nm, stride=st, relu=True)
        if dcn:
            y = deformable_2b(y, p, nm)
        else:
            y = _conv_bn(y, p, 'res%s_branch2b' % nm, 'bn%s_branch2b' % nm, pad=dil, dil=dil, relu=True)
        y = _conv_bn(y, p, 'res%s_branch2c' % nm, 'bn%s_branch2c' % nm)
        x = F.relu(sc + y)
    return x


def dcn_pool(feat, rois, p, sample_per_part=4, trans_std=0.1):
    """offset_t -> FC `offset` -> reshape (-1,2,7,7) -> deformable_roi_pool (SYM_DCN_RELNMS:1073-1080)."""
    feat = np.asarray(feat, np.float32)
    t0, _ = OD.deformable_psroi_pooling(feat, rois, None, 0.0625, feat.shape[1], 1, 7, 7, sample_per_part, 0.0, True)
    off = OR.fc(t0.reshape(t0.shape[0], -1), np.asarray(p['offset_weight']), np.asarray(p['offset_bias']))
    trans = off.reshape(-1, 2, 7, 7).astype(np.float32)
    out, _ = OD.deformable_psroi_pooling(feat, rois, trans, 0.0625, feat.shape[1], 1, 7, 7, sample_per_part, trans_std, False)
    return out, trans


def backbone(data, p, dcn=False, fpn=False):
    """data [B,3,H,W] -> (conv4 [B,1024,h,w], conv5 [B,2048,h,w]); fpn: (res2c, res3b3, res4b22, res5c)."""
    x = _conv_bn(_t(data), p, 'conv1', 'bn_conv1', stride=2, pad=3, relu=True)
    x = F.max_pool2d(x, 3, 2, 0, ceil_mode=True)             # pooling_convention='full'
    ends = {}
    for si, n in enumerate(UNITS):
        stage = si + 2
        if stage == 5:
            c5 = res5(x, p, dcn, fpn)
            return (ends[2], ends[3], x, c5) if fpn else (x, c5)
        for u in range(n):
            nm = _unit_name(stage, u, n)
            stride = 2 if (u == 0 and stage in (3, 4)) else 1     # stride on the first 1x1
            dil = 2 if stage == 5 else 1
            sc = _conv_bn(x, p, 'res%s_branch1' % nm, 'bn%s_branch1' % nm, stride=stride) if u == 0 else x
            y = _conv_bn(x, p, 'res%s_branch2a' % nm, 'bn%s_branch2a' % nm, stride=stride, relu=True)
            y = _conv_bn(y, p, 'res%s_branch2b' % nm, 'bn%s_branch2b' % nm, pad=dil, dil=dil, relu=True)
            y = _conv_bn(y, p, 'res%s_branch2c' % nm, 'bn%s_branch2c' % nm)
            x = F.relu(sc + y)
        ends[stage] = x


def rpn_and_feat(conv4, conv5, p):
    r = F.relu(F.conv2d(conv4, _t(p['rpn_conv_3x3_weight']), _t(p['rpn_conv_3x3_bias']), padding=1))
    cls = F.conv2d(r, _t(p['rpn_cls_score_weight']), _t(p['rpn_cls_score_bias']))
    box = F.conv2d(r, _t(p['rpn_bbox_pred_weight']), _t(p['rpn_bbox_pred_bias']))
    feat = F.relu(F.conv2d(conv5, _t(p['conv_new_1_weight']), _t(p['conv_new_1_bias'])))
    return cls, box, feat


def rpn_softmax(rpn_cls_score):
    """Reshape (0,2,-1,0) + SoftmaxActivation(mode='channel') + reshape back, SYM_REL:218-223."""
    z = np.asarray(rpn_cls_score, dtype=np.float32)
    b, c2, h, w = z.shape
    z = z.reshape(b, 2, c2 // 2 * h, w)
    m = z.max(axis=1, keepdims=True)
    e = np.exp((z - m).astype(np.float64)).astype(np.float32)
    return (e / e.sum(axis=1, keepdims=True)).reshape(b, c2, h, w).astype(np.float32)


def plain_head(roi_pool, p, dtype=np.float32):
    """resnet_v1_101_rcnn.py:125-134: fc_new_1 -> relu -> fc_new_2 -> relu -> cls/bbox."""
    x1 = np.maximum(OR.fc(roi_pool, p['fc_new_1_weight'], p['fc_new_1_bias'], dtype), 0)
    x2 = np.maximum(OR.fc(x1, p['fc_new_2_weight'], p['fc_new_2_bias'], dtype), 0)
    return (OR.fc(x2, p['cls_score_weight'], p['cls_score_bias'], dtype),
            OR.fc(x2, p['bbox_pred_weight'], p['bbox_pred_bias'], dtype), x2)


def detect(data, im_info, p, relation=True, soft=True, nms=0.6, num_classes=81, max_per_image=100,
           scales=(4, 8, 16, 32), ratios=(0.5, 1, 2), pre_nms=6000, post_nms=300, rpn_thresh=0.7):
    """One image through the whole test graph + post-processing (float32 CPU)."""
    pn = {k: (v.numpy() if torch.is_tensor(v) else np.asarray(v)) for k, v in p.items()}
    with torch.no_grad():
        conv4, conv5 = backbone(data, p)
        cls, box, feat = rpn_and_feat(conv4, conv5, p)
    prob = rpn_softmax(cls.numpy())
    rois, _ = OP.proposal(prob, box.numpy(), im_info, 16, scales, ratios, pre_nms, post_nms, rpn_thresh, 0)
    pooled = ORP.roi_pooling(feat.numpy(), rois)
    if relation:
        r = OR.relation_head(pooled, rois, pn, return_intermediates=True)
        cls_score, bbox = r['cls_score'], r['bbox_pred']
    else:
        cls_score, bbox, _ = plain_head(pooled, pn)
    import time
    cls_prob = OPP.softmax_rows(cls_score)
    scores, boxes = OPP.im_detect(rois, cls_prob, bbox, im_info)
    t0 = time.time()
    dets = OPP.detections(scores, boxes, num_classes, 1e-3, nms, soft, max_per_image)
    return dict(rois=rois, cls_prob=cls_prob, bbox_pred=bbox, boxes=boxes, dets=dets, post_seconds=time.time() - t0)
