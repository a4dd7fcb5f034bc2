"""Seeded synthetic inputs shared by the golden generator and the tests
(fixed seeds, tie-free scores, boxes inside a 600x1000 image).

Large tensors (features, weights) are regenerated from the seed instead of being
stored in the fixtures; only the reference-side OUTPUTS are committed as .npz.
"""
import numpy as np

F32 = np.float32
IM_H, IM_W = 600, 1000


def random_boxes(n, seed, im_h=IM_H, im_w=IM_W, min_size=16, max_size=400):
    """n boxes (x1,y1,x2,y2) fp32, integer-free uniform, clipped to the image."""
    rng = np.random.default_rng(seed)
    x1 = rng.uniform(0, im_w - min_size - 1, n)
    y1 = rng.uniform(0, im_h - min_size - 1, n)
    w = rng.uniform(min_size, max_size, n)
    h = rng.uniform(min_size, max_size, n)
    x2 = np.minimum(x1 + w, im_w - 1)
    y2 = np.minimum(y1 + h, im_h - 1)
    return np.stack((x1, y1, x2, y2), axis=1).astype(F32)


def relation_case(n, m, seed, std, feat_dim=1024, dim=(1024, 1024, 1024), fc_dim=16, emb=64,
                  index=1, prefix=''):
    """boxes [n,4], feat [n,feat_dim], relation weights N(0,std) (biases small non-zero so
    bias handling is exercised; the reference initialises them to 0 but trains them)."""
    rng = np.random.default_rng(seed)
    boxes = random_boxes(n, seed + 1000)
    feat = rng.normal(0, 1, (n, feat_dim)).astype(F32)
    p = {}
    def nrm(*shape, s=std):
        return rng.normal(0.0, s, size=shape).astype(F32)
    p['%spair_pos_fc1_%d_weight' % (prefix, index)] = nrm(fc_dim, emb)
    p['%spair_pos_fc1_%d_bias' % (prefix, index)] = nrm(fc_dim, s=std)
    p['%squery_%d_weight' % (prefix, index)] = nrm(dim[0], feat_dim)
    p['%squery_%d_bias' % (prefix, index)] = nrm(dim[0], s=std)
    p['%skey_%d_weight' % (prefix, index)] = nrm(dim[1], feat_dim)
    p['%skey_%d_bias' % (prefix, index)] = nrm(dim[1], s=std)
    p['%slinear_out_%d_weight' % (prefix, index)] = nrm(dim[2], feat_dim, 1, 1)
    p['%slinear_out_%d_bias' % (prefix, index)] = nrm(dim[2], s=std)
    return boxes, feat, p


RELATION_CASES = {
    # name: (n, m, seed, std)
    'rel_n48_m48_std01': (48, 48, 11, 0.01),
    'rel_n40_m32_std05': (40, 32, 12, 0.05),     # keys = first 32 rois (nongt_dim < N)
}


def learn_nms_case(n, num_fg, seed, std=0.05):
    """Inputs of the learn_nms operator (operator_py/learn_nms.py:437-441 argument order)."""
    rng = np.random.default_rng(seed)
    boxes = random_boxes(n, seed + 2000)
    rois = np.hstack((np.zeros((n, 1), F32), boxes)).astype(F32)
    cls_score = rng.normal(0, 2.0, (n, num_fg + 1)).astype(F32)
    bbox_pred = rng.normal(0, 0.1, (n, 8)).astype(F32)
    im_info = np.array([[IM_H, IM_W, 1.0]], dtype=F32)
    feat = np.maximum(rng.normal(0, 1, (n, 1024)), 0).astype(F32)
    def nrm(*shape, s=std):
        return rng.normal(0.0, s, size=shape).astype(F32)
    p = {
        'nms_rank_weight': nrm(128, 1024), 'nms_rank_bias': nrm(128),
        'roi_feat_embedding_weight': nrm(128, 1024), 'roi_feat_embedding_bias': nrm(128),
        'nms_pair_pos_fc1_1_weight': nrm(16, 64), 'nms_pair_pos_fc1_1_bias': nrm(16),
        'nms_query_1_weight': nrm(1024, 128), 'nms_query_1_bias': nrm(1024),
        'nms_key_1_weight': nrm(1024, 128), 'nms_key_1_bias': nrm(1024),
        'nms_linear_out_1_weight': nrm(128, 128, 1, 1), 'nms_linear_out_1_bias': nrm(128),
        'nms_logit_weight': nrm(5, 128), 'nms_logit_bias': np.full(5, -3.0, F32),
    }
    return cls_score, bbox_pred, rois, im_info, feat, p


LEARN_NMS_ARG_ORDER = ['nms_rank_weight', 'nms_rank_bias', 'roi_feat_embedding_weight',
                       'roi_feat_embedding_bias', 'nms_pair_pos_fc1_1_weight',
                       'nms_pair_pos_fc1_1_bias', 'nms_query_1_weight', 'nms_query_1_bias',
                       'nms_key_1_weight', 'nms_key_1_bias', 'nms_linear_out_1_weight',
                       'nms_linear_out_1_bias', 'nms_logit_weight', 'nms_logit_bias']

#: full-size cases (tests/golden/relation_large.npz keeps a SUBSET of query rows: the logits of one N = 1000 case are 64 MB)
RELATION_LARGE_CASES = {
    # name: (n, m, seed, std, rows kept)
    'rel_n300_m300_std01': (300, 300, 13, 0.01, 48),
    'rel_n333_m300_std05': (333, 300, 14, 0.05, 48),      # training shape: 300 proposals + gt rows, keys = first 300
    'rel_n1000_m1000_std02': (1000, 1000, 15, 0.02, 24),  # FPN configuration (TOP_ROIS 1000)
}


def kept_rows(n, k, seed):
    """The query rows stored for a large case: first, last and k-2 seeded random ones (sorted)."""
    rng = np.random.default_rng(seed + 5000)
    r = set([0, n - 1]) | set(int(x) for x in rng.choice(n, k, replace=False))
    return np.array(sorted(r)[:k], dtype=np.int64)


LEARN_NMS_LARGE_CASES = {
    'lnms_n300_c80_f100': (300, 80, 100, 22),               # the benchmark's learn-NMS shape
}

LEARN_NMS_CASES = {
    # name: (n_rois, num_fg_classes, first_n, seed)
    'lnms_n60_c6_f20': (60, 6, 20, 21),
}

#: the learn-NMS head AS THE FPN YAML WORDS IT (experiments/relation_rcnn/cfgs/..._rcnn_fpn_relation_learn_nms_8epoch.yaml:141,166-167:
#: TOP_ROIS 1000, FIRST_N 150, LEARN_NMS_CLASS_SCORE_TH 0.05); stored in tests/golden/learn_nms_fpn.npz
LEARN_NMS_FPN_CASES = {
    # name: (n_rois, num_fg_classes, first_n, seed, class_thresh)
    'lnms_n1000_c80_f150_th05': (1000, 80, 150, 23, 0.05),
}


def learn_nms_fpn_case(n, num_fg, seed):
    """learn_nms_case with every fourth class pushed down by 5 logits: 9 of the 80 class maxima fall below 0.01 and another 9
    between 0.01 and 0.05, so the yaml's 0.05 valid-class rule and the default 0.01 give different class sets."""
    cls_score, bbox_pred, rois, im_info, feat, p = learn_nms_case(n, num_fg, seed)
    cls_score[:, 1 + np.arange(0, num_fg, 4)] -= F32(5.0)
    return cls_score, bbox_pred, rois, im_info, feat, p


def dets_case(n, seed):
    """[n,5] fp32 detections with distinct scores for NMS tests."""
    rng = np.random.default_rng(seed)
    boxes = random_boxes(n, seed + 3000, max_size=300)
    scores = rng.permutation(n).astype(np.float64)
    scores = (scores + rng.uniform(0.1, 0.9, n)) / n       # distinct, in (0,1)
    return np.hstack((boxes, scores[:, None].astype(F32))).astype(F32)


def rpn_case(seed, height=38, width=63, num_anchors=12, score_sigma=2.0, delta_sigma=0.5):
    """cls_prob [1,2A,H,W] (softmax pairs), bbox_pred [1,4A,H,W]; fg scores tie-free."""
    rng = np.random.default_rng(seed)
    logit = rng.normal(0, score_sigma, (1, num_anchors, height, width))
    fg = 1.0 / (1.0 + np.exp(-logit))
    fg = fg.astype(F32)
    flat = fg.ravel()
    # break exact fp32 ties deterministically
    order = np.argsort(flat, kind='stable')
    s = flat[order]
    for _ in range(4):
        dup = np.where(s[1:] <= s[:-1])[0]
        if dup.size == 0:
            break
        for d in dup:
            s[d + 1] = np.nextafter(s[d], F32(2.0), dtype=F32)
    flat[order] = s
    fg = flat.reshape(fg.shape)
    cls_prob = np.concatenate((F32(1) - fg, fg), axis=1).astype(F32)
    deltas = rng.normal(0, delta_sigma, (1, 4 * num_anchors, height, width)).astype(F32)
    im_info = np.array([[IM_H, IM_W, 1.0]], dtype=F32)
    return cls_prob, deltas, im_info


def targets_case(n, g, seed, num_fg=80):
    """Training-target inputs: rois [n,5], gt_boxes [g,5] (class ids 1..num_fg), and head outputs."""
    rng = np.random.default_rng(seed)
    gt = random_boxes(g, seed + 4000, min_size=32)
    gt_cls = rng.integers(1, num_fg + 1, g).astype(F32)
    gt_boxes = np.hstack((gt, gt_cls[:, None])).astype(F32)
    rois = random_boxes(n, seed + 4001)
    # a third of the rois are jittered copies of gt boxes so that positives exist
    k = n // 3
    src = rng.integers(0, g, k)
    rois[:k] = np.clip(gt[src] + rng.normal(0, 6, (k, 4)).astype(F32), 0, [IM_W - 1, IM_H - 1, IM_W - 1, IM_H - 1])
    rois[:k, 2:] = np.maximum(rois[:k, 2:], rois[:k, :2] + 4)
    rois = np.hstack((np.zeros((n, 1), F32), rois)).astype(F32)
    cls_score = rng.normal(0, 2, (n + g, num_fg + 1)).astype(F32)
    bbox_pred = rng.normal(0, 0.5, (n + g, 8)).astype(F32)
    return rois, gt_boxes, cls_score, bbox_pred


def nms_target_case(first_n, num_fg, g, seed):
    """bbox [F,C,4], gt_box [1,G,5], score [F,C] (descending per class, like sorted_score)."""
    rng = np.random.default_rng(seed)
    gt = random_boxes(g, seed + 5000, min_size=40)
    gt_cls = rng.integers(1, min(num_fg, 4) + 1, g).astype(F32)      # few classes -> several gts per class
    gt_box = np.hstack((gt, gt_cls[:, None])).astype(F32)[None]
    bbox = np.zeros((first_n, num_fg, 4), F32)
    for c in range(num_fg):
        b = random_boxes(first_n, seed + 6000 + c)
        src = rng.integers(0, g, first_n // 2)
        b[:first_n // 2] = np.clip(gt[src] + rng.normal(0, 8, (first_n // 2, 4)).astype(F32), 0, [IM_W - 1, IM_H - 1, IM_W - 1, IM_H - 1])
        b[:, 2:] = np.maximum(b[:, 2:], b[:, :2] + 4)
        bbox[:, c] = b[rng.permutation(first_n)]
    score = np.sort(rng.random((first_n, num_fg)).astype(F32), axis=0)[::-1].copy()
    return bbox, gt_box, score


def fpn_proposals(n, seed, im_h=800, im_w=1024):
    """n float32 proposals spanning all four pyramid levels; the first six sit exactly ON the level boundaries
    (sqrt(w*h) = 112, 224, 448 with w = h and with w = 4h)."""
    rng = np.random.default_rng(seed)
    side = np.exp(rng.uniform(np.log(12), np.log(700), n))
    ar = np.exp(rng.uniform(-0.7, 0.7, n))
    w = np.minimum(side * ar, im_w - 2); h = np.minimum(side / ar, im_h - 2)
    x1 = rng.uniform(0, im_w - 1 - w); y1 = rng.uniform(0, im_h - 1 - h)
    b = np.stack([x1, y1, x1 + w, y1 + h], 1).astype(F32)
    for i, s in enumerate((112, 224, 448)):
        b[i] = [10, 20, 10 + s - 1, 20 + s - 1]
        b[3 + i] = [5, 5, 5 + 2 * s - 1, 5 + s / 2 - 1]
    return b


def rpn_gt_boxes(g, seed, im_h=IM_H, im_w=IM_W):
    """g gt boxes [g,5] (x1,y1,x2,y2,cls) float32 with sides in [40, 400]."""
    rng = np.random.default_rng(seed)
    bw, bh = rng.uniform(40, 400, g), rng.uniform(40, 400, g)
    x1, y1 = rng.uniform(0, im_w - 1 - bw), rng.uniform(0, im_h - 1 - bh)
    return np.stack([x1, y1, x1 + bw, y1 + bh, rng.integers(1, 81, g)], 1).astype(F32)


def cocoeval_case(seed=71, n_images=6, n_cats=4):
    """A synthetic detection problem for the COCO bbox evaluation: ground truth with crowd boxes and every area range, detections
    that hit, miss, duplicate and drift, several per (image, category), distinct scores.  -> (gts, dts) lists of dicts in the
    COCO json layout (bbox = [x, y, w, h])."""
    rng = np.random.default_rng(seed)
    gts, dts = [], []
    gid = 1
    for img in range(1, n_images + 1):
        for cat in range(1, n_cats + 1):
            ng = int(rng.integers(0, 5))
            for _ in range(ng):
                side = float(np.exp(rng.uniform(np.log(8), np.log(300))))
                ar = float(np.exp(rng.uniform(-0.6, 0.6)))
                w, h = min(side * ar, 590.0), min(side / ar, 390.0)
                x, y = float(rng.uniform(0, 600 - w)), float(rng.uniform(0, 400 - h))
                crowd = int(rng.random() < 0.12)
                gts.append(dict(id=gid, image_id=img, category_id=cat, bbox=[x, y, w, h], area=w * h * float(rng.uniform(0.5, 1.0)),
                                iscrowd=crowd))
                gid += 1
                for k in range(int(rng.integers(0, 4))):        # detections around this gt: jittered copies
                    j = rng.normal(0, 0.08 + 0.1 * k, 4)
                    dts.append(dict(image_id=img, category_id=cat, score=float(rng.random()),
                                    bbox=[x + j[0] * w, y + j[1] * h, w * float(np.exp(j[2])), h * float(np.exp(j[3]))]))
            for _ in range(int(rng.integers(0, 3))):            # false positives
                w, h = float(rng.uniform(5, 200)), float(rng.uniform(5, 200))
                dts.append(dict(image_id=img, category_id=cat, score=float(rng.random()) * 0.7,
                                bbox=[float(rng.uniform(0, 600 - w)), float(rng.uniform(0, 400 - h)), w, h]))
    for i, d in enumerate(dts):
        d['id'] = i + 1
        d['area'] = d['bbox'][2] * d['bbox'][3]
    return gts, dts
