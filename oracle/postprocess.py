"""Oracle: test-time post-processing (numpy).  TEST INFRASTRUCTURE ONLY.

Follows relation_rcnn/core/tester.py: :148-156 (`im_detect`: bbox_pred + clip_boxes, / scale),
:244-268 (per-class threshold 1e-3, class-agnostic boxes[:, 4:8], `nms` or `soft_nms`),
:270-277 (max_per_image).  tester.py itself has Python-2 prints and cannot be imported; the
pieces it calls (bbox_pred, clip_boxes, nms, soft_nms) are pinned by tests/golden.
"""
import numpy as np

from .boxes import bbox_pred, clip_boxes
from .nms import py_nms, soft_nms


def softmax_rows(z):
    """mx.sym.SoftmaxActivation over the class axis, float32 in / out."""
    z = np.asarray(z, dtype=np.float32)
    e = np.exp((z - z.max(axis=1, keepdims=True)).astype(np.float64))
    return (e / e.sum(axis=1, keepdims=True)).astype(np.float32)


def im_detect(rois, cls_prob, bbox_deltas, im_info):
    """rois [N,5] fp32, cls_prob [N,C] fp32, bbox_deltas [N,4*num_reg] fp32 -> scores, boxes(float64)."""
    info = np.asarray(im_info).reshape(-1, 3)[0]
    boxes = bbox_pred(np.asarray(rois)[:, 1:], bbox_deltas)
    boxes = clip_boxes(boxes, (info[0], info[1]))
    return cls_prob, boxes / info[2]


def detections(scores, boxes, num_classes=81, thresh=1e-3, nms_param=0.6, soft=True, max_per_image=100,
               class_agnostic=True):
    """-> list over classes 1..C-1 of [k,5] arrays (x1,y1,x2,y2,score), after max_per_image."""
    all_boxes = [None] * num_classes
    for j in range(1, num_classes):
        idx = np.where(scores[:, j] > thresh)[0]
        cls_scores = scores[idx, j, np.newaxis]
        cls_boxes = boxes[idx, 4:8] if class_agnostic else boxes[idx, j * 4:(j + 1) * 4]
        cls_dets = np.hstack((cls_boxes, cls_scores))
        if soft:
            all_boxes[j] = soft_nms(cls_dets, nms_param, -1)
        else:
            keep = py_nms(cls_dets, nms_param)
            all_boxes[j] = cls_dets[keep, :] if len(keep) else np.zeros((0, 5))
    if max_per_image > 0:
        image_scores = np.hstack([all_boxes[j][:, -1] for j in range(1, num_classes)])
        if len(image_scores) > max_per_image:
            image_thresh = np.sort(image_scores)[-max_per_image]
            for j in range(1, num_classes):
                keep = np.where(all_boxes[j][:, -1] >= image_thresh)[0]
                all_boxes[j] = all_boxes[j][keep, :]
    return all_boxes[1:]
