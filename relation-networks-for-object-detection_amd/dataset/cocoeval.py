"""COCO bounding-box evaluation (AP / AR over IoU 0.50:0.05:0.95, area ranges, maxDets 1 / 10 / 100) in numpy.

The reference evaluates through its vendored pycocotools (`lib/dataset/pycocotools/cocoeval.py` @ pdollar/coco 3ac47c7,
called from `lib/dataset/coco.py:237-249`); that is a C / Cython extension which is not built here, so the published
algorithm is restated: per (image, category) the detections, sorted by score, are matched greedily to the ground truth
of highest IoU still available at each threshold (crowd boxes may be matched repeatedly and use intersection / detection
area as their IoU; ignored ground truth is tried last), precision is made monotone and sampled at 101 recall points.
PINNED: tests/golden/cocoeval.npz holds the 12 statistics and the full precision / recall arrays produced by the reference's
own `cocoeval.py` (executed by tests/golden/gen_golden.py through lib2to3, its C `mask.iou` replaced by a transcription of
maskApi.c:98-109) on a synthetic problem with crowd boxes; tests/test_dataset.py holds this module to them at 1e-12, next to
hand-made known-answer cases.
"""
import numpy as np


class Params(object):
    def __init__(self):
        self.iouThrs = np.linspace(0.5, 0.95, int(np.round((0.95 - 0.5) / 0.05)) + 1, endpoint=True)
        self.recThrs = np.linspace(0.0, 1.00, int(np.round((1.00 - 0.0) / 0.01)) + 1, endpoint=True)
        self.maxDets = [1, 10, 100]
        self.areaRng = [[0 ** 2, 1e5 ** 2], [0 ** 2, 32 ** 2], [32 ** 2, 96 ** 2], [96 ** 2, 1e5 ** 2]]
        self.areaRngLbl = ['all', 'small', 'medium', 'large']
        self.useCats = 1


def bbox_iou(dt, gt, iscrowd):
    """dt [D,4], gt [G,4] in (x, y, w, h); crowd ground truth: intersection over DETECTION area (maskApi bbIou)."""
    dt, gt = np.asarray(dt, np.float64).reshape(-1, 4), np.asarray(gt, np.float64).reshape(-1, 4)
    if len(dt) == 0 or len(gt) == 0:
        return np.zeros((len(dt), len(gt)))
    da, ga = dt[:, 2] * dt[:, 3], gt[:, 2] * gt[:, 3]
    w = np.minimum(dt[:, None, 0] + dt[:, None, 2], gt[None, :, 0] + gt[None, :, 2]) - np.maximum(dt[:, None, 0], gt[None, :, 0])
    h = np.minimum(dt[:, None, 1] + dt[:, None, 3], gt[None, :, 1] + gt[None, :, 3]) - np.maximum(dt[:, None, 1], gt[None, :, 1])
    inter = np.clip(w, 0, None) * np.clip(h, 0, None)
    crowd = np.asarray(iscrowd, bool)[None, :]
    union = np.where(crowd, da[:, None], da[:, None] + ga[None, :] - inter)
    return inter / np.maximum(union, 1e-300)


class COCOeval(object):
    """gts / dts: lists of dicts with image_id, category_id, bbox [x,y,w,h], (area, iscrowd, id) / score."""

    def __init__(self, gts, dts, img_ids=None, cat_ids=None):
        self.params = Params()
        self.gts, self.dts = {}, {}
        for i, g in enumerate(gts):
            g = dict(g)
            g.setdefault('iscrowd', 0); g.setdefault('id', i + 1)
            g.setdefault('area', g['bbox'][2] * g['bbox'][3])
            g['ignore'] = int(g.get('ignore', 0) or g['iscrowd'])
            self.gts.setdefault((g['image_id'], g['category_id']), []).append(g)
        for i, d in enumerate(dts):
            d = dict(d)
            d.setdefault('id', i + 1)
            d.setdefault('area', d['bbox'][2] * d['bbox'][3])
            self.dts.setdefault((d['image_id'], d['category_id']), []).append(d)
        self.img_ids = sorted(img_ids if img_ids is not None else {k[0] for k in list(self.gts) + list(self.dts)})
        self.cat_ids = sorted(cat_ids if cat_ids is not None else {k[1] for k in list(self.gts) + list(self.dts)})
        self.eval_imgs, self.eval, self.stats = {}, None, None

    def _evaluate_img(self, img, cat, area, max_det):
        p = self.params
        gt, dt = self.gts.get((img, cat), []), self.dts.get((img, cat), [])
        if not gt and not dt:
            return None
        gig = np.array([g['ignore'] or g['area'] < area[0] or g['area'] > area[1] for g in gt], dtype=bool)
        gorder = np.argsort(gig, kind='mergesort')                       # not-ignored first
        gt = [gt[i] for i in gorder]
        gig = gig[gorder]
        dorder = np.argsort([-d['score'] for d in dt], kind='mergesort')[:max_det]
        dt = [dt[i] for i in dorder]
        crowd = [int(g['iscrowd']) for g in gt]
        ious = bbox_iou([d['bbox'] for d in dt], [g['bbox'] for g in gt], crowd)
        T, G, D = len(p.iouThrs), len(gt), len(dt)
        gtm, dtm, dtig = np.zeros((T, G)), np.zeros((T, D)), np.zeros((T, D), dtype=bool)
        for ti, t in enumerate(p.iouThrs):
            for di in range(D):
                iou, m = min(t, 1 - 1e-10), -1
                for gi in range(G):
                    if gtm[ti, gi] > 0 and not crowd[gi]:
                        continue
                    if m > -1 and not gig[m] and gig[gi]:
                        break                                           # only ignored gt left, a regular match exists
                    if ious[di, gi] < iou:
                        continue
                    iou, m = ious[di, gi], gi
                if m == -1:
                    continue
                dtig[ti, di] = gig[m]
                dtm[ti, di] = gt[m]['id']
                gtm[ti, m] = dt[di]['id']
        out_of_range = np.array([d['area'] < area[0] or d['area'] > area[1] for d in dt], dtype=bool).reshape(1, D)
        dtig = np.logical_or(dtig, np.logical_and(dtm == 0, np.repeat(out_of_range, T, 0)))
        return dict(dtMatches=dtm, dtScores=np.array([d['score'] for d in dt]), gtIgnore=gig, dtIgnore=dtig)

    def evaluate(self):
        p = self.params
        max_det = p.maxDets[-1]
        self.eval_imgs = {(c, ai, i): self._evaluate_img(i, c, area, max_det)
                          for c in self.cat_ids for ai, area in enumerate(p.areaRng) for i in self.img_ids}

    def accumulate(self):
        p = self.params
        T, R, K, A, M = len(p.iouThrs), len(p.recThrs), len(self.cat_ids), len(p.areaRng), len(p.maxDets)
        precision, recall = -np.ones((T, R, K, A, M)), -np.ones((T, K, A, M))
        for k, c in enumerate(self.cat_ids):
            for a in range(A):
                E = [self.eval_imgs[(c, a, i)] for i in self.img_ids]
                E = [e for e in E if e is not None]
                if not E:
                    continue
                for m, max_det in enumerate(p.maxDets):
                    scores = np.concatenate([e['dtScores'][:max_det] for e in E])
                    inds = np.argsort(-scores, kind='mergesort')
                    dtm = np.concatenate([e['dtMatches'][:, :max_det] for e in E], axis=1)[:, inds]
                    dtig = np.concatenate([e['dtIgnore'][:, :max_det] for e in E], axis=1)[:, inds]
                    npig = int(np.count_nonzero(np.concatenate([e['gtIgnore'] for e in E]) == 0))
                    if npig == 0:
                        continue
                    tps = np.logical_and(dtm, np.logical_not(dtig))
                    fps = np.logical_and(np.logical_not(dtm), np.logical_not(dtig))
                    tp_sum, fp_sum = np.cumsum(tps, axis=1).astype(float), np.cumsum(fps, axis=1).astype(float)
                    for t in range(T):
                        tp, fp = tp_sum[t], fp_sum[t]
                        nd = len(tp)
                        rc = tp / npig
                        pr = tp / (fp + tp + np.spacing(1))
                        recall[t, k, a, m] = rc[-1] if nd else 0
                        pr = pr.tolist()
                        for i in range(nd - 1, 0, -1):
                            if pr[i] > pr[i - 1]:
                                pr[i - 1] = pr[i]
                        q = np.zeros(R)
                        idx = np.searchsorted(rc, p.recThrs, side='left')
                        for ri, pi in enumerate(idx):
                            if pi < nd:
                                q[ri] = pr[pi]
                        precision[t, :, k, a, m] = q
        self.eval = dict(precision=precision, recall=recall, counts=[T, R, K, A, M])

    def _summarize(self, ap=1, iou_thr=None, area='all', max_dets=100):
        p = self.params
        a = p.areaRngLbl.index(area)
        m = p.maxDets.index(max_dets)
        s = self.eval['precision'] if ap else self.eval['recall']
        if iou_thr is not None:
            s = s[np.where(np.isclose(p.iouThrs, iou_thr))[0]]
        s = s[..., a, m] if ap else s[..., a, m]
        v = s[s > -1]
        return float(np.mean(v)) if v.size else -1.0

    def summarize(self):
        """The 12 COCO detection metrics, in pycocotools' order."""
        st = [self._summarize(1), self._summarize(1, iou_thr=.5), self._summarize(1, iou_thr=.75),
              self._summarize(1, area='small'), self._summarize(1, area='medium'), self._summarize(1, area='large'),
              self._summarize(0, max_dets=1), self._summarize(0, max_dets=10), self._summarize(0, max_dets=100),
              self._summarize(0, area='small'), self._summarize(0, area='medium'), self._summarize(0, area='large')]
        self.stats = np.array(st)
        return self.stats
