"""Oracle: learned duplicate-removal (learn-NMS) head, inference (numpy).
TEST INFRASTRUCTURE ONLY.

Follows relation_rcnn/operator_py/learn_nms.py: :238-401 LearnNmsOperator.forward,
:175-217 refine_bbox_nd, :129-140 extract_rank_embedding_nd, :142-172
extract_multi_position_matrix_nd, :21-43 pairwise embedding, :45-127
nms_attention_nd; merge over thresholds as symbols/..._learn_nms.py:553-560.
Wiring pinned by tests/golden (the reference operator executed on the numpy MXNet
stand-in); MXNet per-op semantics restated.
"""
import numpy as np

from .relation import position_matrix, position_embedding, relation_module, fc, cr

F32 = np.float32


def refine_boxes(rois_xyxy, deltas, im_info=None, means=None, stds=None, dtype=F32):
    """learn_nms.py:175-217.  NOTE the centre is 0.5*(x1+x2) and the half extent
    0.5*(w'-1) here (:180-181,198-199) -- not the x1+0.5*(w-1) form of bbox_transform.
    rois [N,4], deltas [N, 4k] -> [N, 4, k]; clipped to [0, im_w-1] x [0, im_h-1]."""
    b = np.asarray(rois_xyxy, dtype=dtype)
    d = np.asarray(deltas, dtype=dtype).reshape(b.shape[0], -1, 4)
    one, half = dtype(1.0), dtype(0.5)
    w = b[:, 2:3] - b[:, 0:1] + one
    h = b[:, 3:4] - b[:, 1:2] + one
    cx = half * (b[:, 0:1] + b[:, 2:3])
    cy = half * (b[:, 1:2] + b[:, 3:4])
    dx, dy, dw, dh = (d[:, :, i] for i in range(4))
    if means is not None and stds is not None:
        dx = dx * dtype(stds[0]) + dtype(means[0]); dy = dy * dtype(stds[1]) + dtype(means[1])
        dw = dw * dtype(stds[2]) + dtype(means[2]); dh = dh * dtype(stds[3]) + dtype(means[3])
    rcx = cx + w * dx
    rcy = cy + h * dy
    rw = w * cr(np.exp, dw)
    rh = h * cr(np.exp, dh)
    wo = half * (rw - one)
    ho = half * (rh - one)
    out = np.stack((rcx - wo, rcy - ho, rcx + wo, rcy + ho), axis=1).astype(dtype)  # [N,4,k]
    if im_info is not None:
        info = np.asarray(im_info, dtype=dtype).reshape(-1, 3)[0]
        lim = np.array([info[1] - one, info[0] - one, info[1] - one, info[0] - one], dtype=dtype)
        out = np.maximum(np.minimum(out, lim.reshape(1, 4, 1)), dtype(0))
    return out


def rank_embedding(rank_dim, feat_dim=1024, wave_length=1000, dtype=F32):
    """learn_nms.py:129-140: arg = r / wave^((2/feat_dim) k), [sin | cos]."""
    r = np.arange(0, rank_dim).astype(dtype)[:, None]
    k = np.arange(0, feat_dim // 2).astype(dtype)
    dim = np.power(dtype(wave_length), dtype(2.0 / feat_dim) * k).astype(dtype)[None, :]
    div = r / dim
    return np.concatenate((cr(np.sin, div), cr(np.cos, div)), axis=1).astype(dtype)


def learn_nms(cls_score, bbox_pred, rois, im_info, fc_all_2_relu, params, num_fg_classes=80,
              first_n=100, num_thresh=5, class_thresh=0.01, nongt_dim=None, means=None,
              stds=None, class_agnostic=True, dtype=F32, return_intermediates=False):
    """-> nms_multi_score [first_n, C, T], sorted_bbox [first_n, C, 4], sorted_score [first_n, C]."""
    assert class_agnostic, "shipped cfgs are class agnostic (CLASS_AGNOSTIC: true)"
    p = params
    cls_score = np.asarray(cls_score, dtype=dtype)
    bbox_pred = np.asarray(bbox_pred, dtype=dtype)
    rois = np.asarray(rois, dtype=dtype)
    feat = np.asarray(fc_all_2_relu, dtype=dtype)
    if nongt_dim is not None:                                        # :265-267,282-283
        cls_score, bbox_pred = cls_score[:nongt_dim], bbox_pred[:nongt_dim]
    boxes = rois[:, 1:]
    if nongt_dim is not None:
        boxes = boxes[:nongt_dim]
    refined = refine_boxes(boxes, bbox_pred[:, 4:], im_info, means, stds, dtype)   # [N,4,1]
    z = cls_score.astype(np.float64)
    ez = np.exp(z - z.max(axis=1, keepdims=True))
    prob = (ez / ez.sum(axis=1, keepdims=True)).astype(dtype)[:, 1:]               # [N, C]
    rank = np.argsort(-prob, axis=0, kind='stable')[:first_n]                      # [first_n, C]
    sorted_score = np.take_along_axis(prob, rank, axis=0)
    max_per_class = sorted_score.max(axis=0)
    thr = np.minimum(class_thresh, max_per_class.max())                            # :295-296
    valid = np.where(max_per_class >= thr)[0]
    sorted_bbox = refined[rank][:, :, :, 0]                                        # [first_n, C, 4]
    rank_feat = fc(rank_embedding(first_n, 1024, 1000, dtype), p['nms_rank_weight'],
                   p['nms_rank_bias'], dtype)                                      # [first_n,128]
    roi_emb = fc(feat, p['roi_feat_embedding_weight'], p['roi_feat_embedding_bias'], dtype)
    multi = np.zeros((first_n, num_fg_classes, num_thresh), dtype=dtype)
    inter = {}
    for c in valid:
        x = (roi_emb[rank[:, c]] + rank_feat).astype(dtype)                        # [first_n,128]
        pm = position_matrix(sorted_bbox[:, c, :], first_n, dtype)
        pe = position_embedding(pm, 64, 1000, dtype)
        att = relation_module(x, pe, p, index=1, nongt_dim=first_n, fc_dim=16, feat_dim=128,
                              dim=(1024, 1024, 128), group=16, dtype=dtype, prefix='nms_')
        zr = np.maximum(x + att, dtype(0))
        logit = fc(zr, p['nms_logit_weight'], p['nms_logit_bias'], dtype)          # [first_n,T]
        cond = (dtype(1) / (dtype(1) + cr(np.exp, -logit))).astype(dtype)
        multi[:, c, :] = sorted_score[:, c:c + 1] * cond
        if return_intermediates:
            inter[int(c)] = dict(x=x, attention=att, logit=logit)
    if return_intermediates:
        return multi, sorted_bbox, sorted_score, dict(valid=valid, rank=rank, per_class=inter)
    return multi, sorted_bbox, sorted_score


def merge_thresholds(nms_multi_score, merge_method=-1):
    """symbols/..._learn_nms.py:553-560: -1 mean, -2 max, k>=0 slice."""
    if merge_method == -1:
        return nms_multi_score.astype(np.float64).mean(axis=2).astype(nms_multi_score.dtype)
    if merge_method == -2:
        return nms_multi_score.max(axis=2)
    return nms_multi_score[:, :, merge_method]
