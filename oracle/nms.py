"""Oracle: the three NMS flavours of the reference (numpy).  TEST INFRASTRUCTURE ONLY.

  * `gpu_nms`  -- lib/nms/gpu_nms.pyx:18-33 + lib/nms/nms_kernel.cu:24-32 (fp32 IoU,
                  suppress when IoU > thresh) + :118-140 (greedy scan in score order).
                  PINNED (round 4): lib/nms/nms_kernel.cu compiled unedited for gfx950
                  (oracle/build_ref.py) and run on an MI355X gives the keep lists stored in
                  tests/golden/ref_cuda.npz (duplicated boxes, 64 / 65-box block edges, one box);
                  tests/test_oracle_refcuda.py holds `nms_sorted_f32` to them, and
                  tests/test_gpu_refcuda.py the product's `_nms` to the reference's `_nms`.
  * `py_nms`   -- lib/nms/nms.py:45-82 (keep while ovr <= thresh).   Pinned.
  * `soft_nms` -- lib/nms/nms.py:85-141 (gaussian rescoring, re-sort each step). Pinned.
"""
import numpy as np

F32 = np.float32


def argsort_desc(scores):
    """Descending order exactly as the reference spells it: `argsort()[::-1]`
    (proposal.py:140, gpu_nms.pyx:28, nms.py:62).  numpy's default sort is not
    stable, so the order of exactly tied scores is unspecified in the reference;
    test inputs are tie-free, and the HIP path documents its own tie rule
    (ties -> higher original index first, i.e. the reversed stable sort)."""
    return np.argsort(scores, kind='stable')[::-1]


def iou_f32(a, b):
    """devIoU of nms_kernel.cu:24-32 evaluated in float32, one rounding per op
    (no FMA contraction): a (4,), b (n, 4)."""
    a = np.asarray(a, dtype=F32)
    b = np.asarray(b, dtype=F32)
    one, zero = F32(1), F32(0)
    left = np.maximum(a[0], b[:, 0]); right = np.minimum(a[2], b[:, 2])
    top = np.maximum(a[1], b[:, 1]); bottom = np.minimum(a[3], b[:, 3])
    width = np.maximum((right - left) + one, zero)
    height = np.maximum((bottom - top) + one, zero)
    inter = width * height
    sa = ((a[2] - a[0]) + one) * ((a[3] - a[1]) + one)
    sb = ((b[:, 2] - b[:, 0]) + one) * ((b[:, 3] - b[:, 1]) + one)
    return inter / ((sa + sb) - inter)


def nms_sorted_f32(boxes_sorted, thresh, max_keep=-1):
    """Greedy scan over boxes already sorted by score (nms_kernel.cu:118-140):
    box i is kept unless an earlier kept box j has IoU(j, i) > thresh (:71).
    Returns positions (ascending) into the sorted array."""
    boxes_sorted = np.asarray(boxes_sorted, dtype=F32)
    n = boxes_sorted.shape[0]
    removed = np.zeros(n, dtype=bool)
    keep = []
    t = F32(thresh)
    for i in range(n):
        if removed[i]:
            continue
        keep.append(i)
        if max_keep > 0 and len(keep) >= max_keep:
            break
        if i + 1 < n:
            ov = iou_f32(boxes_sorted[i, :4], boxes_sorted[i + 1:, :4])
            removed[i + 1:] |= ov > t
    return np.asarray(keep, dtype=np.int64)


def gpu_nms(dets, thresh):
    """gpu_nms.pyx:18-33: sort desc, run `_nms`, map back to unsorted indices."""
    dets = np.asarray(dets, dtype=F32)
    if dets.shape[0] == 0:
        return []
    order = argsort_desc(dets[:, 4])
    keep = nms_sorted_f32(dets[order, :4], thresh)
    return list(order[keep])


def py_nms(dets, thresh):
    """nms.py:45-82, arithmetic in the dtype of `dets`."""
    dets = np.asarray(dets)
    if dets.shape[0] == 0:
        return []
    x1, y1, x2, y2, scores = (dets[:, i] for i in range(5))
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = argsort_desc(scores)
    keep = []
    while order.size > 0:
        i = order[0]
        keep.append(i)
        r = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[r]) - np.maximum(x1[i], x1[r]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[r]) - np.maximum(y1[i], y1[r]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[r] - inter)
        order = r[np.where(ovr <= thresh)[0]]
    return keep


def soft_nms(dets, thresh, max_dets=-1):
    """Gaussian soft-NMS, nms.py:96-141 with `rescore` :85-93
    (score *= exp(-ovr^2 / thresh)); returns the re-scored rows in pick order."""
    dets = np.array(dets, copy=True)
    if dets.shape[0] == 0:
        return np.zeros((0, 5))
    x1, y1, x2, y2 = (dets[:, i] for i in range(4))
    scores = dets[:, 4]
    areas = (x2 - x1 + 1) * (y2 - y1 + 1)
    order = argsort_desc(scores)
    scores = scores[order]
    if max_dets == -1:
        max_dets = order.size
    keep = []
    while order.size > 0 and len(keep) < max_dets:
        i = order[0]
        dets[i, 4] = scores[0]
        r = order[1:]
        w = np.maximum(0.0, np.minimum(x2[i], x2[r]) - np.maximum(x1[i], x1[r]) + 1)
        h = np.maximum(0.0, np.minimum(y2[i], y2[r]) - np.maximum(y1[i], y1[r]) + 1)
        inter = w * h
        ovr = inter / (areas[i] + areas[r] - inter)
        scores = scores[1:] * np.exp(-ovr ** 2 / thresh)
        tmp = argsort_desc(scores)
        order = r[tmp]
        scores = scores[tmp]
        keep.append(i)
    return dets[np.asarray(keep, dtype=np.intp), :]
