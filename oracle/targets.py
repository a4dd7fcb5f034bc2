"""Oracle: training-target operators (numpy).  TEST INFRASTRUCTURE ONLY.

  proposal_target  relation_rcnn/operator_py/proposal_target.py:44-93 with BATCH_ROIS = -1 (every
                   shipped e2e cfg): rois + gt boxes -> core/rcnn.py:288-325 `sample_rois_v2`
                   (no sampling, hence deterministic) -> lib/bbox/bbox_regression.py:120-140.
                   proposal_target.py / bbox_regression.py have Python-2 prints: the glue and
                   `expand_bbox_regression_targets` are restated (PARITY UNPINNED); sample_rois_v2 and
                   bbox_transform / bbox_overlaps_py are pinned by tests/golden.
  box_annotator_ohem  operator_py/box_annotator_ohem.py:26-53 (pinned: the reference file runs on the
                   MXNet stand-in).
  nms_multi_target operator_py/nms_multi_target.py:24-74 (pinned likewise).
"""
import numpy as np

from .boxes import bbox_overlaps, bbox_transform

F32 = np.float32


def bbox_transform_f32(ex, gt):
    """bbox_transform.py:74-100 evaluated in float32 (sample_rois_v2 passes float32 arrays, so numpy
    keeps float32; the log is pinned as correctly rounded)."""
    ex = np.asarray(ex, dtype=F32); gt = np.asarray(gt, dtype=F32)
    one, half = F32(1.0), F32(0.5)
    ew = ex[:, 2] - ex[:, 0] + one; eh = ex[:, 3] - ex[:, 1] + one
    ecx = ex[:, 0] + half * (ew - one); ecy = ex[:, 1] + half * (eh - one)
    gw = gt[:, 2] - gt[:, 0] + one; gh = gt[:, 3] - gt[:, 1] + one
    gcx = gt[:, 0] + half * (gw - one); gcy = gt[:, 1] + half * (gh - one)
    # (ew + 1e-14) stays ew in float32
    dx = (gcx - ecx) / (ew + F32(1e-14)); dy = (gcy - ecy) / (eh + F32(1e-14))
    dw = np.log((gw / ew).astype(np.float64)).astype(F32); dh = np.log((gh / eh).astype(np.float64)).astype(F32)
    return np.vstack((dx, dy, dw, dh)).T


def expand_bbox_regression_targets(data, num_classes, class_agnostic=True, bbox_weights=(1.0, 1.0, 1.0, 1.0)):
    """bbox_regression.py:120-140."""
    classes = data[:, 0]
    if class_agnostic:
        num_classes = 2
    t = np.zeros((classes.size, 4 * num_classes), dtype=F32)
    w = np.zeros(t.shape, dtype=F32)
    for i in np.where(classes > 0)[0]:
        cls = classes[i]
        start = 4 if class_agnostic else int(4 * cls)
        t[i, start:start + 4] = data[i, 1:]
        w[i, start:start + 4] = bbox_weights
    return t, w


def proposal_target(rois, gt_boxes, num_classes=81, class_agnostic=True, bg_thresh_hi=0.5,
                    means=(0.0, 0.0, 0.0, 0.0), stds=(0.1, 0.1, 0.2, 0.2), normalize=True):
    """rois [N,5] fp32, gt_boxes [G,5] (x1,y1,x2,y2,cls) fp32 -> rois_output [N+G,5], label [N+G],
    bbox_target [N+G,8], bbox_weight [N+G,8]   (BATCH_ROIS = -1)."""
    rois = np.asarray(rois, dtype=F32); gt = np.asarray(gt_boxes, dtype=F32)
    all_rois = np.vstack((rois, np.hstack((np.zeros((gt.shape[0], 1), F32), gt[:, :-1]))))   # proposal_target.py:64-67
    assert np.all(all_rois[:, 0] == 0), 'Only single item batches are supported'
    ov = bbox_overlaps(all_rois[:, 1:].astype(np.float64), gt[:, :4].astype(np.float64))     # rcnn.py:303
    assign = ov.argmax(axis=1)
    mx = ov.max(axis=1)
    labels = gt[assign, 4].copy()
    labels[mx < bg_thresh_hi] = 0                                                             # :309-310
    targets = bbox_transform_f32(all_rois[:, 1:], gt[assign, :4])                             # :316
    if normalize:
        targets = (targets - np.array(means)) / np.array(stds)                                # float64
    data = np.hstack((labels[:, None], targets))
    bt, bw = expand_bbox_regression_targets(data, num_classes, class_agnostic)
    return all_rois, labels, bt, bw


def smooth_l1(x, scalar=1.0):
    """mx.nd.smooth_l1: 0.5 (s x)^2 if |x| < 1/s^2 else |x| - 0.5/s^2 (float32)."""
    x = np.asarray(x, dtype=F32)
    s2 = F32(scalar * scalar)
    return np.where(np.abs(x) < F32(1.0) / s2, F32(0.5) * s2 * x * x, np.abs(x) - F32(0.5) / s2).astype(F32)


def box_annotator_ohem(cls_score, bbox_pred, labels, bbox_targets, bbox_weights, roi_per_img=128):
    """box_annotator_ohem.py:26-53 -> labels_ohem [N], bbox_weights_ohem [N, 4*num_reg]."""
    z = np.asarray(cls_score, dtype=F32)
    e = np.exp((z - z.max(axis=1, keepdims=True)).astype(np.float64))
    prob = (e / e.sum(axis=1, keepdims=True)).astype(F32) + F32(1e-14)
    lab = np.asarray(labels).astype(np.int64)
    loss_cls = -np.log(prob[np.arange(len(lab)), lab].astype(np.float64)).astype(F32)
    lb = (np.asarray(bbox_weights, F32) * smooth_l1(np.asarray(bbox_pred, F32) - np.asarray(bbox_targets, F32))).astype(np.float64).sum(axis=1).astype(F32)
    order = np.argsort(loss_cls + lb, kind='stable')[::-1]
    lo = np.array(labels, dtype=F32, copy=True)
    wo = np.array(bbox_weights, dtype=F32, copy=True)
    lo[order[roi_per_img:]] = -1
    wo[order[roi_per_img:]] = 0
    return lo, wo, loss_cls + lb


def nms_multi_target(bbox, gt_box, score, target_thresh=(0.5, 0.6, 0.7, 0.8, 0.9)):
    """nms_multi_target.py:24-74: bbox [F, C, 4], gt_box [1, G, 5], score [F, C] -> [F, C, T]."""
    bbox = np.asarray(bbox); gt_box = np.asarray(gt_box); score = np.asarray(score)
    F, C = bbox.shape[:2]
    T = len(target_thresh)
    out = np.zeros((F, C, T), dtype=F32)
    for c in range(C):
        vg = gt_box[0, gt_box[0, :, -1].astype(np.int32) == c + 1, :]
        if len(vg) == 0:
            continue
        ov = bbox_overlaps(bbox[:, c, :].astype(np.float64), vg[:, :-1].astype(np.float64))
        eye = np.eye(len(vg))
        amax = np.argmax(ov, axis=1)
        for t, th in enumerate(target_thresh):
            mask = ov > th
            valid = np.where(mask)[0]
            os_ = np.tile(score[:, c:c + 1], (1, len(vg))) * mask * eye[amax]
            best = np.argmax(os_, axis=0)
            out[np.intersect1d(best, valid), c, t] = 1
    return out
