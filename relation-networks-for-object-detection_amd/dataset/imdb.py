"""Image database base class: `lib/dataset/imdb.py` (roidb = list of per-image dicts with `boxes`, `gt_classes`,
`gt_overlaps`, `max_classes`, `max_overlaps`, `flipped`, `is_gt`, `image`, `height`, `width`).

The precomputed-proposal format is the reference's (`core/tester.py:118-126`, `imdb.py:103-138`): a pickle holding one
float array [n, 5] (x1, y1, x2, y2, score, ORIGINAL image scale) per image, in image-set order, at
`<rpn_path>/rpn_data/<name>_rpn.pkl`."""
import os
import pickle

import numpy as np


class IMDB(object):
    def __init__(self, name, image_set, root_path, dataset_path, result_path=None, rpn_path=None):
        self.name = name + '_' + image_set
        self.image_set = image_set
        self.root_path = root_path
        self.data_path = dataset_path
        self._result_path = result_path
        self._rpn_path = rpn_path
        self.classes = []
        self.num_classes = 0
        self.image_set_index = []
        self.num_images = 0

    def image_path_from_index(self, index):
        raise NotImplementedError

    def gt_roidb(self):
        raise NotImplementedError

    def evaluate_detections(self, detections):
        raise NotImplementedError

    @property
    def cache_path(self):
        p = os.path.join(self.root_path, 'cache')
        os.makedirs(p, exist_ok=True)
        return p

    @property
    def result_path(self):
        p = self._result_path if self._result_path else self.cache_path
        os.makedirs(p, exist_ok=True)
        return p

    @property
    def rpn_path(self):
        return self._rpn_path if self._rpn_path else self.root_path

    def image_path_at(self, index):
        return self.image_path_from_index(self.image_set_index[index])

    # ---- precomputed proposals (imdb.py:103-188) ---------------------------------------------------------
    def rpn_file(self, full=False):
        return os.path.join(self.rpn_path, 'rpn_data', self.name + ('_full_rpn.pkl' if full else '_rpn.pkl'))

    def save_rpn_data(self, box_list, full=False):
        f = self.rpn_file(full)
        os.makedirs(os.path.dirname(f), exist_ok=True)
        with open(f, 'wb') as fid:
            pickle.dump(box_list, fid, pickle.HIGHEST_PROTOCOL)
        return f

    def load_rpn_data(self, full=False):
        f = self.rpn_file(full)
        assert os.path.exists(f), 'rpn data not found at {}'.format(f)
        with open(f, 'rb') as fid:
            try:
                return pickle.load(fid)
            except UnicodeDecodeError:                      # a pickle written by the Python-2 reference
                fid.seek(0)
                return pickle.load(fid, encoding='latin1')

    def load_rpn_roidb(self, gt_roidb, top_roi=-1):
        box_list = self.load_rpn_data()
        if top_roi != -1:
            box_list = [boxes[:top_roi, :] for boxes in box_list]
        return self.create_roidb_from_box_list(box_list, gt_roidb)

    def rpn_roidb(self, gt_roidb, append_gt=False, top_roi=-1):
        rpn = self.load_rpn_roidb(gt_roidb, top_roi)
        return IMDB.merge_roidbs(rpn, gt_roidb) if append_gt else rpn

    # per-roi fields of a roidb record and how two records of one image are joined (row-wise / element-wise)
    _ROW_FIELDS = ('boxes', 'gt_overlaps')
    _VEC_FIELDS = ('gt_classes', 'max_classes', 'max_overlaps', 'is_gt')

    def _class_overlaps(self, boxes, gt, overlaps_fn):
        """[n, num_classes] float32: for every box the IoU with its best ground-truth box, stored in that box's class
        column (zero elsewhere, and zero rows for boxes that touch no ground truth)."""
        table = np.zeros((len(boxes), self.num_classes), dtype=np.float32)
        if gt is None or gt['boxes'].size == 0 or len(boxes) == 0:
            return table
        iou = np.asarray(overlaps_fn(boxes.astype(np.float64), gt['boxes'].astype(np.float64)))
        best = iou.argmax(axis=1)
        val = iou[np.arange(len(boxes)), best]
        hit = val > 0
        table[np.flatnonzero(hit), gt['gt_classes'][best[hit]]] = val[hit]
        return table

    def create_roidb_from_box_list(self, box_list, gt_roidb, overlaps_fn=None):
        """Proposal records for `lib/dataset/imdb.py:140-188`'s consumers (same keys, dtypes and values).
        `overlaps_fn(boxes f64 [N,4], gt f64 [K,4]) -> [N,K]` defaults to the device twin of `bbox_overlaps_cython`
        (relnet_amd.bbox)."""
        if len(box_list) != self.num_images:
            raise AssertionError('number of boxes matrix must match number of images')
        if overlaps_fn is None:
            from ..bbox import bbox_overlaps as overlaps_fn

        def record(dets, gt):
            boxes = dets[:, :4]                                 # a fifth column is the proposal score
            ov = self._class_overlaps(boxes, gt, overlaps_fn)
            return {'image': gt['image'], 'height': gt['height'], 'width': gt['width'], 'boxes': boxes,
                    'gt_classes': np.zeros(len(boxes), dtype=np.int32), 'gt_overlaps': ov,
                    'max_classes': ov.argmax(axis=1), 'max_overlaps': ov.max(axis=1),
                    'flipped': False, 'is_gt': np.zeros(len(boxes))}
        return [record(d, g) for d, g in zip(box_list, gt_roidb)]

    def append_flipped_images(self, roidb):
        """Doubles the roidb with mirrored entries (`imdb.py:219-255`): x' = width - 1 - x with the two corners swapped;
        the pixels themselves are mirrored when the image is loaded.  Every other field is shared with the original."""
        if len(roidb) != self.num_images:
            raise AssertionError('roidb does not cover the image set')

        def mirrored(rec):
            b = rec['boxes'].copy()
            b[:, [0, 2]] = rec['width'] - 1 - rec['boxes'][:, [2, 0]]
            if (b[:, 2] < b[:, 0]).any():
                raise AssertionError('box with x2 < x1 in %s' % rec['image'])
            return dict(rec, boxes=b, flipped=True)
        roidb.extend([mirrored(r) for r in roidb[:self.num_images]])
        self.image_set_index = list(self.image_set_index) + list(self.image_set_index)
        return roidb

    @staticmethod
    def merge_roidbs(a, b):
        """Joins, image by image, the rois of `b` below those of `a` (proposals + ground truth, `imdb.py:382-400`);
        `a` is updated in place and returned."""
        if len(a) != len(b):
            raise AssertionError('roidbs of different image sets')
        for ra, rb in zip(a, b):
            ra.update({k: np.concatenate((ra[k], rb[k]), axis=0) for k in IMDB._ROW_FIELDS + IMDB._VEC_FIELDS})
        return a
