"""Oracle: RPN proposal operator (numpy).  TEST INFRASTRUCTURE ONLY.

Follows relation_rcnn/operator_py/proposal.py:51-168 of the reference.  That file
has Python-2 print statements and imports mxnet + the CUDA NMS, so it cannot be
imported here: the glue below is a restatement (PARITY UNPINNED) built on pieces
that ARE pinned against the reference's python (anchors, bbox_pred, clip_boxes).
"""
import numpy as np

from .boxes import generate_anchors, bbox_pred, clip_boxes
from .nms import argsort_desc, nms_sorted_f32

F32 = np.float32


def shifted_anchors(height, width, feat_stride, base_anchors):
    """All anchors in (y, x, a) order: proposal.py:88-104."""
    sx = np.arange(0, width) * feat_stride
    sy = np.arange(0, height) * feat_stride
    sx, sy = np.meshgrid(sx, sy)
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).T
    a = base_anchors.shape[0]
    k = shifts.shape[0]
    return (base_anchors.reshape(1, a, 4) + shifts.reshape(k, 1, 4)).reshape(k * a, 4)


def proposal(cls_prob, bbox_deltas, im_info, feat_stride=16, scales=(4, 8, 16, 32),
             ratios=(0.5, 1, 2), pre_nms_top_n=6000, post_nms_top_n=300,
             threshold=0.7, min_size=0, return_debug=False, rng=None):
    """cls_prob [1, 2A, H, W] fp32, bbox_deltas [1, 4A, H, W] fp32, im_info [1,3]
    -> rois [post_nms_top_n, 5] fp32, scores [post_nms_top_n, 1] fp32.

    proposal.py:58 batch check, :75 fg scores = channels [A:], :85 cropped grid
    int(im/stride), :113-123 (h, w, a) flattening, :126-129 decode (float64) +
    clip, :133-135 min-size filter, :140-144 descending order + top-N,
    :149 cast to float32, :150-158 NMS / truncate / random pad, :163-168 output."""
    cls_prob = np.asarray(cls_prob)
    bbox_deltas = np.asarray(bbox_deltas)
    if cls_prob.shape[0] > 1:
        raise ValueError("Sorry, multiple images each device is not implemented")
    base = generate_anchors(base_size=feat_stride, ratios=ratios, scales=scales)
    A = base.shape[0]
    info = np.asarray(im_info).reshape(-1, 3)[0]
    height, width = int(info[0] / feat_stride), int(info[1] / feat_stride)
    scores = cls_prob[:, A:, :height, :width]
    deltas = bbox_deltas[:, :, :height, :width]
    anchors = shifted_anchors(height, width, feat_stride, base)
    deltas = deltas.transpose(0, 2, 3, 1).reshape(-1, 4)
    scores = scores.transpose(0, 2, 3, 1).reshape(-1, 1)
    props = bbox_pred(anchors, deltas)
    props = clip_boxes(props, info[:2])
    ms = min_size * info[2]
    ws = props[:, 2] - props[:, 0] + 1
    hs = props[:, 3] - props[:, 1] + 1
    valid = np.where((ws >= ms) & (hs >= ms))[0]
    props = props[valid]
    scores = scores[valid]
    order = argsort_desc(scores.ravel())
    if pre_nms_top_n > 0:
        order = order[:pre_nms_topn_clamp(pre_nms_top_n, order.size)]
    props = props[order]
    scores = scores[order]
    det = np.hstack((props, scores)).astype(F32)
    # gpu_nms re-sorts by score (gpu_nms.pyx:28); `det` is already descending and
    # tie-free, so the scan order is the row order.
    keep = nms_sorted_f32(det[:, :4], threshold)
    n_kept = len(keep)
    if post_nms_top_n > 0:
        keep = keep[:post_nms_top_n]
    if len(keep) < post_nms_top_n:
        rng = rng or np.random
        pad = rng.choice(keep, size=post_nms_top_n - len(keep))
        keep = np.hstack((keep, pad))
    rois = np.hstack((np.zeros((len(keep), 1), dtype=F32), props[keep].astype(F32)))
    out_scores = scores[keep].astype(F32)
    if return_debug:
        return rois, out_scores, dict(order=valid[order], keep=keep, n_kept=n_kept, det=det)
    return rois, out_scores


def pre_nms_topn_clamp(n, size):
    return min(int(n), int(size))
