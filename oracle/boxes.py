"""Oracle: anchors and box arithmetic (numpy).  TEST INFRASTRUCTURE ONLY.

Follows lib/rpn/generate_anchor.py:22-86 and lib/bbox/bbox_transform.py:45-140,
lib/bbox/bbox.pyx:15-55 of the reference.  Pinned against those files by
tests/golden/gen_golden.py (they are importable under Python 3 with shims).
"""
import numpy as np


def _whctrs(a):
    w = a[2] - a[0] + 1.0
    h = a[3] - a[1] + 1.0
    return w, h, a[0] + 0.5 * (w - 1.0), a[1] + 0.5 * (h - 1.0)


def _mk(ws, hs, cx, cy):
    ws = np.asarray(ws, dtype=np.float64).reshape(-1, 1)
    hs = np.asarray(hs, dtype=np.float64).reshape(-1, 1)
    return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1),
                      cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """Base anchors around the (0,0,base-1,base-1) window, ratio-major/scale-minor.
    generate_anchor.py:22-34 (driver), :61-72 (ratio enum), :75-86 (scale enum)."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)
    base = np.array([0, 0, base_size - 1, base_size - 1], dtype=np.float64)
    w, h, cx, cy = _whctrs(base)
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    ratio_anchors = _mk(ws, hs, cx, cy)
    out = []
    for ra in ratio_anchors:
        w, h, cx, cy = _whctrs(ra)
        out.append(_mk(w * scales, h * scales, cx, cy))
    return np.vstack(out)


def bbox_pred(boxes, deltas):
    """Decode deltas (N, 4k) against boxes (N, 4) in float64.
    bbox_transform.py:103-140 (`nonlinear_pred`, exported as `bbox_pred`)."""
    boxes = np.asarray(boxes)
    deltas = np.asarray(deltas)
    if boxes.shape[0] == 0:
        return np.zeros((0, deltas.shape[1]))
    boxes = boxes.astype(np.float64, copy=False)
    w = boxes[:, 2] - boxes[:, 0] + 1.0
    h = boxes[:, 3] - boxes[:, 1] + 1.0
    cx = boxes[:, 0] + 0.5 * (w - 1.0)
    cy = boxes[:, 1] + 0.5 * (h - 1.0)
    dx, dy, dw, dh = (deltas[:, i::4] for i in range(4))
    pcx = dx * w[:, None] + cx[:, None]
    pcy = dy * h[:, None] + cy[:, None]
    # the reference evaluates np.exp in the dtype of the deltas (float32 from
    # `.asnumpy()`); pinned here as the correctly rounded value in that dtype.
    pw = np.exp(dw.astype(np.float64)).astype(dw.dtype) * w[:, None]
    ph = np.exp(dh.astype(np.float64)).astype(dh.dtype) * h[:, None]
    out = np.zeros(deltas.shape, dtype=np.result_type(pcx.dtype, np.float64))
    out[:, 0::4] = pcx - 0.5 * (pw - 1.0)
    out[:, 1::4] = pcy - 0.5 * (ph - 1.0)
    out[:, 2::4] = pcx + 0.5 * (pw - 1.0)
    out[:, 3::4] = pcy + 0.5 * (ph - 1.0)
    return out


def clip_boxes(boxes, im_shape):
    """Clip to [0, w-1] x [0, h-1]; im_shape = (h, w).  bbox_transform.py:45-60."""
    boxes = np.array(boxes, copy=True)
    boxes[:, 0::4] = np.maximum(np.minimum(boxes[:, 0::4], im_shape[1] - 1), 0)
    boxes[:, 1::4] = np.maximum(np.minimum(boxes[:, 1::4], im_shape[0] - 1), 0)
    boxes[:, 2::4] = np.maximum(np.minimum(boxes[:, 2::4], im_shape[1] - 1), 0)
    boxes[:, 3::4] = np.maximum(np.minimum(boxes[:, 3::4], im_shape[0] - 1), 0)
    return boxes


def bbox_overlaps(boxes, query):
    """float64 IoU matrix (N, K) with +1 extents; zero when disjoint.
    bbox.pyx:33-55 (same maths as bbox_transform.py:22-42)."""
    b = np.asarray(boxes, dtype=np.float64)
    q = np.asarray(query, dtype=np.float64)
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0]) + 1
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1]) + 1
    ab = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    aq = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    inter = iw * ih
    ua = ab[:, None] + aq[None, :] - inter
    return np.where((iw > 0) & (ih > 0), inter / ua, 0.0)


def bbox_transform(ex, gt):
    """Regression targets ex -> gt.  bbox_transform.py:74-100."""
    ex = np.asarray(ex, dtype=np.float64)
    gt = np.asarray(gt, dtype=np.float64)
    ew = ex[:, 2] - ex[:, 0] + 1.0
    eh = ex[:, 3] - ex[:, 1] + 1.0
    ecx = ex[:, 0] + 0.5 * (ew - 1.0)
    ecy = ex[:, 1] + 0.5 * (eh - 1.0)
    gw = gt[:, 2] - gt[:, 0] + 1.0
    gh = gt[:, 3] - gt[:, 1] + 1.0
    gcx = gt[:, 0] + 0.5 * (gw - 1.0)
    gcy = gt[:, 1] + 0.5 * (gh - 1.0)
    return np.vstack(((gcx - ecx) / (ew + 1e-14), (gcy - ecy) / (eh + 1e-14),
                      np.log(gw / ew), np.log(gh / eh))).T
