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

    def create_roidb_from_box_list(self, box_list, gt_roidb, overlaps_fn=None):
        """imdb.py:140-188.  `overlaps_fn(boxes f64 [N,4], gt f64 [K,4]) -> [N,K]` defaults to the device twin of
        `bbox_overlaps_cython` (relnet_amd.bbox)."""
        assert len(box_list) == self.num_images, 'number of boxes matrix must match number of images'
        if overlaps_fn is None:
            from ..bbox import bbox_overlaps as overlaps_fn
        roidb = []
        for i in range(self.num_images):
            rec = dict(image=gt_roidb[i]['image'], height=gt_roidb[i]['height'], width=gt_roidb[i]['width'])
            boxes = box_list[i]
            if boxes.shape[1] == 5:
                boxes = boxes[:, :4]
            n = boxes.shape[0]
            overlaps = np.zeros((n, self.num_classes), dtype=np.float32)
            if gt_roidb is not None and gt_roidb[i]['boxes'].size > 0:
                gt_boxes, gt_classes = gt_roidb[i]['boxes'], gt_roidb[i]['gt_classes']
                gt_ov = np.asarray(overlaps_fn(boxes.astype(np.float64), gt_boxes.astype(np.float64)))
                argmaxes, maxes = gt_ov.argmax(axis=1), gt_ov.max(axis=1)
                idx = np.where(maxes > 0)[0]
                overlaps[idx, gt_classes[argmaxes[idx]]] = maxes[idx]
            rec.update(boxes=boxes, gt_classes=np.zeros((n,), dtype=np.int32), gt_overlaps=overlaps,
                       max_classes=overlaps.argmax(axis=1), max_overlaps=overlaps.max(axis=1), flipped=False,
                       is_gt=np.zeros(n))
            roidb.append(rec)
        return roidb

    def append_flipped_images(self, roidb):
        """imdb.py:219-255: horizontally mirrored copy of every entry (the pixels are flipped when the image is loaded)."""
        assert self.num_images == len(roidb)
        for i in range(self.num_images):
            r = roidb[i]
            boxes = r['boxes'].copy()
            oldx1, oldx2 = boxes[:, 0].copy(), boxes[:, 2].copy()
            boxes[:, 0] = r['width'] - oldx2 - 1
            boxes[:, 2] = r['width'] - oldx1 - 1
            assert (boxes[:, 2] >= boxes[:, 0]).all()
            roidb.append(dict(image=r['image'], height=r['height'], width=r['width'], boxes=boxes, gt_classes=r['gt_classes'],
                              gt_overlaps=r['gt_overlaps'], max_classes=r['max_classes'], max_overlaps=r['max_overlaps'],
                              flipped=True, is_gt=r['is_gt']))
        self.image_set_index = self.image_set_index * 2
        return roidb

    @staticmethod
    def merge_roidbs(a, b):
        """imdb.py:382-400: concatenate the boxes of two roidbs of the same images (proposals + ground truth)."""
        assert len(a) == len(b)
        for i in range(len(a)):
            a[i]['boxes'] = np.vstack((a[i]['boxes'], b[i]['boxes']))
            a[i]['gt_classes'] = np.hstack((a[i]['gt_classes'], b[i]['gt_classes']))
            a[i]['gt_overlaps'] = np.vstack((a[i]['gt_overlaps'], b[i]['gt_overlaps']))
            a[i]['max_classes'] = np.hstack((a[i]['max_classes'], b[i]['max_classes']))
            a[i]['max_overlaps'] = np.hstack((a[i]['max_overlaps'], b[i]['max_overlaps']))
            a[i]['is_gt'] = np.hstack((a[i]['is_gt'], b[i]['is_gt']))
        return a
