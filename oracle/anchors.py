"""Oracle: RPN anchor labels / regression targets (numpy).  TEST INFRASTRUCTURE ONLY.

Follows lib/rpn/rpn.py:80-244 (`assign_anchor`) with lib/bbox/bbox.pyx:33-55 (overlaps, float64) and
lib/bbox/bbox_transform.py:74-100 (targets).  PINNED: with `sampler='numpy'` the random fg / bg sub-sampling is
`np.random.RandomState(seed).choice(...)` in the reference's call order, and tests/test_oracle_golden.py holds the result
bit-for-bit to tests/golden/rpn_targets.npz (the output of the reference's own function, gen_golden.py).  `sampler='hash'`
swaps ONLY the random subset for the device kernel's definition (csrc/targets.hip: keep the anchors with the largest
32-bit hash of (seed, image, anchor)) -- any uniformly random subset is what the reference asks for (:189-204).
"""
import numpy as np

from .boxes import generate_anchors


def anchor_key(seed, b, idx):
    """The kernel's 32-bit hash of (seed, image, anchor index in (y, x, a) order); idx: int array."""
    m = np.uint64(0xffffffff)
    seed = np.uint64(seed)
    x = (idx.astype(np.uint64) * np.uint64(0x9E3779B1)) & m
    x ^= seed & m
    x ^= (np.uint64(b) * np.uint64(0x85EBCA77)) & m
    x ^= ((seed >> np.uint64(32)) * np.uint64(0xC2B2AE3D)) & m
    x ^= x >> np.uint64(16); x = (x * np.uint64(0x7feb352d)) & m
    x ^= x >> np.uint64(15); x = (x * np.uint64(0x846ca68b)) & m
    x ^= x >> np.uint64(16)
    return x


def overlaps64(boxes, query):
    """bbox.pyx:33-55, float64, +1 extents, 0 when disjoint."""
    b, q = np.asarray(boxes, np.float64), np.asarray(query, np.float64)
    iw = np.minimum(b[:, None, 2], q[None, :, 2]) - np.maximum(b[:, None, 0], q[None, :, 0]) + 1
    ih = np.minimum(b[:, None, 3], q[None, :, 3]) - np.maximum(b[:, None, 1], q[None, :, 1]) + 1
    ba = (b[:, 2] - b[:, 0] + 1) * (b[:, 3] - b[:, 1] + 1)
    qa = (q[:, 2] - q[:, 0] + 1) * (q[:, 3] - q[:, 1] + 1)
    inter = iw * ih
    ov = inter / (ba[:, None] + qa[None, :] - inter)
    return np.where((iw > 0) & (ih > 0), ov, 0.0)


def assign_anchor(feat_hw, gt_boxes, im_hw, feat_stride=16, scales=(4, 8, 16, 32), ratios=(0.5, 1, 2), allowed_border=0,
                  rpn_batch_size=256, fg_fraction=0.5, negative_overlap=0.3, positive_overlap=0.7, clobber_positives=False,
                  sampler='numpy', seed=0, image_index=0, return_all=False):
    """-> label [A*h*w] ((a, y, x) order), bbox_target [4A, h, w], bbox_weight [4A, h, w] float32
    (+ the labels before sub-sampling with return_all)."""
    gt_boxes = np.asarray(gt_boxes, np.float32)
    base = generate_anchors(feat_stride, ratios, scales).astype(np.float64)
    A = base.shape[0]
    fh, fw = feat_hw
    sx, sy = np.meshgrid(np.arange(fw) * feat_stride, np.arange(fh) * feat_stride)                # :127-131
    shifts = np.vstack((sx.ravel(), sy.ravel(), sx.ravel(), sy.ravel())).transpose()
    K = shifts.shape[0]
    all_anchors = (base.reshape((1, A, 4)) + shifts.reshape((1, K, 4)).transpose((1, 0, 2))).reshape((K * A, 4))
    total = K * A
    inside = np.where((all_anchors[:, 0] >= -allowed_border) & (all_anchors[:, 1] >= -allowed_border) &
                      (all_anchors[:, 2] < im_hw[1] + allowed_border) & (all_anchors[:, 3] < im_hw[0] + allowed_border))[0]
    anchors = all_anchors[inside]
    labels = np.full(len(inside), -1, np.float32)
    argmax = None
    if gt_boxes.size > 0:
        ov = overlaps64(anchors, gt_boxes[:, :4])
        argmax = ov.argmax(axis=1)                                                                  # :166
        mx = ov[np.arange(len(inside)), argmax]
        gt_max = ov[ov.argmax(axis=0), np.arange(ov.shape[1])]
        gt_arg = np.where(ov == gt_max)[0]                                                          # :170 every tie
        if not clobber_positives:
            labels[mx < negative_overlap] = 0
        labels[gt_arg] = 1
        labels[mx >= positive_overlap] = 1
        if clobber_positives:
            labels[mx < negative_overlap] = 0
    else:
        labels[:] = 0
    labels_all = labels.copy()
    num_fg = int(fg_fraction * rpn_batch_size)
    rng = np.random.RandomState(seed) if sampler == 'numpy' else None
    keys = None if sampler == 'numpy' else ((anchor_key(seed, image_index, inside).astype(np.uint64) << np.uint64(32)) | inside.astype(np.uint64))

    def disable(cands, n_drop):
        if sampler == 'numpy':
            return rng.choice(cands, size=n_drop, replace=False)                                    # npr.choice, :191,200
        return cands[np.argsort(keys[cands], kind='stable')[:n_drop]]                               # the n_drop smallest keys

    fg = np.where(labels == 1)[0]
    if len(fg) > num_fg:
        labels[disable(fg, len(fg) - num_fg)] = -1
    num_bg = rpn_batch_size - int(np.sum(labels == 1))
    bg = np.where(labels == 0)[0]
    if len(bg) > num_bg:
        labels[disable(bg, len(bg) - max(num_bg, 0))] = -1
    targets = np.zeros((len(inside), 4), np.float32)
    if gt_boxes.size > 0:                                                                           # bbox_transform.py:74-100
        g = gt_boxes[argmax, :4].astype(np.float64)
        ew, eh = anchors[:, 2] - anchors[:, 0] + 1.0, anchors[:, 3] - anchors[:, 1] + 1.0
        ecx, ecy = anchors[:, 0] + 0.5 * (ew - 1.0), anchors[:, 1] + 0.5 * (eh - 1.0)
        gw, gh = g[:, 2] - g[:, 0] + 1.0, g[:, 3] - g[:, 1] + 1.0
        gcx, gcy = g[:, 0] + 0.5 * (gw - 1.0), g[:, 1] + 0.5 * (gh - 1.0)
        targets[:] = np.vstack(((gcx - ecx) / (ew + 1e-14), (gcy - ecy) / (eh + 1e-14), np.log(gw / ew), np.log(gh / eh))).transpose()
    weights = np.zeros((len(inside), 4), np.float32)
    weights[labels == 1, :] = 1.0

    def unmap(data, fill):
        out = np.full((total,) + data.shape[1:], fill, np.float32)
        out[inside] = data
        return out

    def lay(l):
        return unmap(l, -1).reshape((fh, fw, A)).transpose(2, 0, 1).reshape(-1)

    T = unmap(targets, 0).reshape((fh, fw, A * 4)).transpose(2, 0, 1)
    W = unmap(weights, 0).reshape((fh, fw, A * 4)).transpose(2, 0, 1)
    if return_all:
        return lay(labels), T, W, lay(labels_all)
    return lay(labels), T, W
