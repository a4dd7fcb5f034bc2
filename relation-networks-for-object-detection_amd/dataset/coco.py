"""COCO image database: `lib/dataset/coco.py:60-330` without pycocotools -- the annotation JSON is indexed with the
standard library, ground-truth records are built exactly as `_load_coco_annotation` (:130-183) does, results are written
in the COCO results format (`_write_coco_results`, :196-233) and scored by dataset/cocoeval.py."""
import json
import os
import pickle

import numpy as np

from .imdb import IMDB
from .cocoeval import COCOeval


class coco(IMDB):
    VIEW_MAP = {'minival2014': 'val2014', 'valminusminival2014': 'val2014', 'test-dev2015': 'test2015'}

    def __init__(self, image_set, root_path, data_path, result_path=None, rpn_path=None, image_ext='.jpg'):
        super(coco, self).__init__('COCO', image_set, root_path, data_path, result_path, rpn_path)
        with open(self._get_ann_file()) as f:
            ds = json.load(f)
        self._images = {im['id']: im for im in ds['images']}
        self._anns = {}
        for a in ds.get('annotations', []):
            self._anns.setdefault(a['image_id'], []).append(a)
        cats = sorted(ds.get('categories', []), key=lambda c: c['id'])
        names = [c['name'] for c in cats]
        self.classes = ['__background__'] + names
        self.num_classes = len(self.classes)
        self._class_to_ind = dict(zip(self.classes, range(self.num_classes)))
        self._class_to_coco_ind = dict(zip(names, [c['id'] for c in cats]))
        self._coco_ind_to_class_ind = {self._class_to_coco_ind[c]: self._class_to_ind[c] for c in names}
        self.image_set_index = sorted(self._images)
        self.num_images = len(self.image_set_index)
        self.data_name = self.VIEW_MAP.get(image_set, image_set)
        self.image_ext = image_ext

    def _get_ann_file(self):
        prefix = 'instances' if 'test' not in self.image_set else 'image_info'
        return os.path.join(self.data_path, 'annotations', prefix + '_' + self.image_set + '.json')

    def image_path_from_index(self, index):
        im = self._images[index]
        if 'file_name' in im and os.path.exists(os.path.join(self.data_path, 'images', self.data_name, im['file_name'])):
            return os.path.join(self.data_path, 'images', self.data_name, im['file_name'])
        return os.path.join(self.data_path, 'images', self.data_name, 'COCO_%s_%012d%s' % (self.data_name, index, self.image_ext))

    def gt_roidb(self):
        cache_file = os.path.join(self.cache_path, self.name + '_gt_roidb.pkl')
        if os.path.exists(cache_file):
            with open(cache_file, 'rb') as fid:
                return pickle.load(fid)
        roidb = [self._load_coco_annotation(i) for i in self.image_set_index]
        with open(cache_file, 'wb') as fid:
            pickle.dump(roidb, fid, pickle.HIGHEST_PROTOCOL)
        return roidb

    def _load_coco_annotation(self, index):
        im = self._images[index]
        width, height = im['width'], im['height']
        objs = []
        for obj in self._anns.get(index, []):
            if obj.get('iscrowd', 0):                      # getAnnIds(iscrowd=False), coco.py:145
                continue
            x, y, w, h = obj['bbox']
            x1, y1 = np.max((0, x)), np.max((0, y))
            x2 = np.min((width - 1, x1 + np.max((0, w - 1))))
            y2 = np.min((height - 1, y1 + np.max((0, h - 1))))
            if obj['area'] > 0 and x2 >= x1 and y2 >= y1:
                objs.append((obj, [x1, y1, x2, y2]))
        n = len(objs)
        boxes = np.zeros((n, 4), dtype=np.uint16)
        gt_classes = np.zeros((n,), dtype=np.int32)
        overlaps = np.zeros((n, self.num_classes), dtype=np.float32)
        for ix, (obj, clean) in enumerate(objs):
            cls = self._coco_ind_to_class_ind[obj['category_id']]
            boxes[ix, :] = clean
            gt_classes[ix] = cls
            overlaps[ix, cls] = 1.0
        return dict(image=self.image_path_from_index(index), height=height, width=width, boxes=boxes, gt_classes=gt_classes,
                    gt_overlaps=overlaps, max_classes=overlaps.argmax(axis=1) if n else np.zeros((0,), np.int64),
                    max_overlaps=overlaps.max(axis=1) if n else np.zeros((0,), np.float32), flipped=False, is_gt=np.ones(n))

    # ---- evaluation ------------------------------------------------------------------------------------------
    def results_list(self, all_boxes):
        """all_boxes[cls][image] = [k,5] (x1,y1,x2,y2,score) -> COCO results records (coco.py:30-57)."""
        res = []
        for cls_ind, cls in enumerate(self.classes):
            if cls == '__background__':
                continue
            cat_id = self._class_to_coco_ind[cls]
            for im_ind, index in enumerate(self.image_set_index):
                dets = np.asarray(all_boxes[cls_ind][im_ind], dtype=np.float64)
                if len(dets) == 0:
                    continue
                xs, ys = dets[:, 0], dets[:, 1]
                ws, hs = dets[:, 2] - xs + 1, dets[:, 3] - ys + 1
                res.extend({'image_id': index, 'category_id': cat_id, 'bbox': [float(xs[k]), float(ys[k]), float(ws[k]), float(hs[k])],
                            'score': float(dets[k, -1])} for k in range(dets.shape[0]))
        return res

    def evaluate_detections(self, detections, ann_type='bbox'):
        """Write `results/detections_<set>_results.json` and, unless this is a test set, score it; returns
        (info_str, stats[12])."""
        assert ann_type == 'bbox'
        res_folder = os.path.join(self.result_path, 'results')
        os.makedirs(res_folder, exist_ok=True)
        res_file = os.path.join(res_folder, 'detections_%s_results.json' % self.image_set)
        results = self.results_list(detections)
        with open(res_file, 'w') as f:
            json.dump(results, f, sort_keys=True, indent=4)
        if 'test' in self.image_set:
            return 'results written to %s\n' % res_file, None
        gts = [a for anns in self._anns.values() for a in anns]
        ev = COCOeval(gts, results, img_ids=self.image_set_index, cat_ids=sorted(self._coco_ind_to_class_ind))
        ev.evaluate(); ev.accumulate()
        stats = ev.summarize()
        info = self._detection_metrics(ev)
        return info, stats

    def _detection_metrics(self, ev):
        """coco.py:251-293: mean and per-category AP over IoU 0.50:0.95, area 'all', 100 detections."""
        prec = ev.eval['precision'][:, :, :, 0, 2]
        v = prec[prec > -1]
        info = '~~~~ Mean and per-category AP @ IoU=[0.50,0.95] ~~~~\n%-15s %5.1f\n' % ('all', 100 * (v.mean() if v.size else -1))
        for k, cat in enumerate(ev.cat_ids):
            pk = prec[:, :, k]
            vk = pk[pk > -1]
            info += '%-15s %5.1f\n' % (self.classes[self._coco_ind_to_class_ind[cat]], 100 * (vk.mean() if vk.size else -1))
        names = ['AP', 'AP50', 'AP75', 'APs', 'APm', 'APl', 'AR1', 'AR10', 'AR100', 'ARs', 'ARm', 'ARl']
        info += '~~~~ Summary metrics ~~~~\n' + ''.join('%-6s %.3f\n' % (n, s) for n, s in zip(names, ev.stats))
        return info
