"""Data side on the CPU: COCO image database without pycocotools, proposal pickles, image
pre-processing, loaders, and the numpy bbox evaluator against hand-computed answers."""
import json
import os
import pickle

import numpy as np
import pytest

import relnet_amd  # noqa: F401
from relnet_amd import config as C
from relnet_amd.dataset import coco as coco_mod, cocoeval, image as IMG, loader as LD
from oracle import boxes as OB


def make_dataset(root, n_images=6, seed=0, image_set='val2014', degenerate=True):
    """A miniature COCO tree: PNG images, 3 categories with non-contiguous ids, crowd / degenerate annotations."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    data_path = os.path.join(root, 'coco')
    os.makedirs(os.path.join(data_path, 'annotations'), exist_ok=True)
    os.makedirs(os.path.join(data_path, 'images', image_set), exist_ok=True)
    cats = [{'id': 1, 'name': 'person'}, {'id': 3, 'name': 'car'}, {'id': 7, 'name': 'train'}]
    images, anns, aid = [], [], 1
    for i in range(n_images):
        w, h = (200, 150) if i % 3 else (140, 210)
        iid = 100 + 7 * i
        fn = 'COCO_%s_%012d.png' % (image_set, iid)
        Image.fromarray(rng.integers(0, 255, (h, w, 3), dtype=np.uint8)).save(os.path.join(data_path, 'images', image_set, fn))
        images.append({'id': iid, 'width': w, 'height': h, 'file_name': fn})
        for k in range(3):
            bw, bh = rng.uniform(20, 90), rng.uniform(20, 90)
            x, y = rng.uniform(0, w - bw - 1), rng.uniform(0, h - bh - 1)
            anns.append({'id': aid, 'image_id': iid, 'category_id': cats[(i + k) % 3]['id'], 'bbox': [x, y, bw, bh], 'area': bw * bh, 'iscrowd': 0})
            aid += 1
        anns.append({'id': aid, 'image_id': iid, 'category_id': 1, 'bbox': [5, 5, 60, 60], 'area': 3600, 'iscrowd': 1}); aid += 1
        if degenerate:
            anns.append({'id': aid, 'image_id': iid, 'category_id': 3, 'bbox': [10, 10, 0, 0], 'area': 0, 'iscrowd': 0}); aid += 1
            anns.append({'id': aid, 'image_id': iid, 'category_id': 7, 'bbox': [w - 10.5, h - 8.5, 40, 40], 'area': 1600, 'iscrowd': 0}); aid += 1
    with open(os.path.join(data_path, 'annotations', 'instances_%s.json' % image_set), 'w') as f:
        json.dump({'images': images, 'annotations': anns, 'categories': cats}, f)
    return coco_mod('val2014', root, data_path, image_ext='.png')


def test_coco_roidb_without_pycocotools(tmp_path):
    db = make_dataset(str(tmp_path))
    assert db.classes == ['__background__', 'person', 'car', 'train'] and db.num_images == 6 and db.name == 'COCO_val2014'
    assert db._coco_ind_to_class_ind == {1: 1, 3: 2, 7: 3}
    roidb = db.gt_roidb()
    r = roidb[0]
    assert os.path.exists(r['image']) and (r['height'], r['width']) == (210, 140)
    assert r['boxes'].dtype == np.uint16 and r['boxes'].shape == (4, 4)          # 3 regular + the clipped corner box; crowd, zero-area gone
    assert tuple(r['boxes'][3]) == (129, 201, 139, 209)                          # x1 = uint16(129.5), clipped to (w-1, h-1)
    assert r['gt_classes'].tolist()[3] == 3 and (r['max_overlaps'] == 1).all() and (r['is_gt'] == 1).all()
    assert np.array_equal(r['gt_overlaps'][np.arange(4), r['gt_classes']], np.ones(4, np.float32))
    assert os.path.exists(os.path.join(db.cache_path, 'COCO_val2014_gt_roidb.pkl'))
    again = db.gt_roidb()                                                        # served from the cache pickle
    assert all(np.array_equal(a['boxes'], b['boxes']) for a, b in zip(roidb, again))
    flipped = db.append_flipped_images([dict(x) for x in roidb])
    assert len(flipped) == 12 and db.image_set_index[6:] == db.image_set_index[:6]
    f0 = flipped[6]
    assert f0['flipped'] and np.array_equal(f0['boxes'][:, 0], 140 - roidb[0]['boxes'][:, 2] - 1)


def test_proposal_pickle_round_trip_and_rpn_roidb(tmp_path):
    db = make_dataset(str(tmp_path))
    gt = db.gt_roidb()
    rng = np.random.default_rng(3)
    box_list = []
    for r in gt:
        n = 40
        x1 = rng.uniform(0, r['width'] - 30, n); y1 = rng.uniform(0, r['height'] - 30, n)
        b = np.stack([x1, y1, x1 + rng.uniform(8, 29, n), y1 + rng.uniform(8, 29, n), rng.uniform(0, 1, n)], 1).astype(np.float32)
        b[0, :4] = r['boxes'][0]                                                  # one proposal exactly on a gt box
        box_list.append(b)
    f = db.save_rpn_data(box_list)
    assert f.endswith(os.path.join('rpn_data', 'COCO_val2014_rpn.pkl'))
    with open(f, 'rb') as fid:
        assert isinstance(pickle.load(fid), list)                                 # the reference's format: a plain list of arrays
    back = db.load_rpn_data()
    assert all(np.array_equal(a, b) for a, b in zip(back, box_list))
    roidb = db.create_roidb_from_box_list([b[:25] for b in back], gt, overlaps_fn=OB.bbox_overlaps)
    r = roidb[0]
    assert r['boxes'].shape == (25, 4) and r['max_overlaps'][0] == 1.0 and r['max_classes'][0] == gt[0]['gt_classes'][0]
    ov = OB.bbox_overlaps(r['boxes'].astype(np.float64), gt[0]['boxes'].astype(np.float64))
    assert np.allclose(r['max_overlaps'], ov.max(1)) and (r['is_gt'] == 0).all() and (r['max_classes'][r['max_overlaps'] == 0] == 0).all()
    merged = coco_mod.merge_roidbs(roidb, gt)
    assert merged[0]['boxes'].shape == (29, 4) and merged[0]['is_gt'].sum() == 4


def test_image_preprocessing_rules():
    cfg = C.experiment('rcnn_end2end_relation_8epoch')
    im = np.zeros((150, 200, 3), np.float32); im[..., 0] = 10; im[..., 1] = 20; im[..., 2] = 30          # B, G, R planes
    out, s = IMG.resize(im, 600, 1000, stride=0)
    assert s == 4.0 and out.shape == (600, 800, 3)
    out, s = IMG.resize(np.zeros((100, 400, 3), np.float32), 600, 1000, stride=32)                       # long side capped, padded to 32
    assert s == 2.5 and out.shape == (256, 1024, 3)
    t = IMG.transform(im, np.array([1.0, 2.0, 3.0]))
    assert t.shape == (1, 3, 150, 200) and (t[0, 0] == 27).all() and (t[0, 1] == 18).all() and (t[0, 2] == 9).all()   # R-3, G-2, B-1
    a, b = np.ones((1, 3, 4, 6), np.float32), np.ones((1, 3, 5, 2), np.float32)
    v = IMG.tensor_vstack([a, b])
    assert v.shape == (2, 3, 5, 6) and v[0, :, 4].sum() == 0 and v[1, :, :, 2:].sum() == 0
    assert cfg.SCALES[0] == (600, 1000)


def test_loaders(tmp_path):
    db = make_dataset(str(tmp_path))
    cfg = C.experiment('rcnn_end2end_relation_8epoch')
    cfg.SCALES[0] = (120, 200)
    roidb = db.gt_roidb()
    tl = LD.TestLoader(roidb, cfg, batch_size=1, has_rpn=True)
    batches = list(tl)
    assert len(batches) == 6 and [b['index'][0] for b in batches] == list(range(6))
    b0 = batches[0]
    assert b0['data'].shape[:2] == (1, 3) and min(b0['data'].shape[2:]) == 120 and 'proposals' not in b0
    assert np.isclose(float(b0['im_info'][0, 2]), 120.0 / 140.0)
    al = LD.AnchorLoader(roidb, cfg, batch_size=2, shuffle=True, aspect_grouping=True, seed=5)
    seen = 0
    for b in al:
        H, W = b['data'].shape[2:]
        fh, fw = LD._conv4_size(H), LD._conv4_size(W)
        assert b['label'].shape == (2, 12 * fh * fw) and b['bbox_target'].shape == (2, 48, fh, fw)
        assert b['gt_boxes'].shape[0] == 2 and b['gt_boxes'].shape[2] == 5 and int(b['num_gt'].min()) == 4
        assert set(np.unique(b['label'].numpy())) <= {-1.0, 0.0, 1.0}
        seen += 1
    assert seen == 3
    bd = next(iter(LD.AnchorLoader(roidb, cfg, batch_size=2, device_targets=True)))       # targets left to the device kernel
    assert set(bd) == {'data', 'im_info', 'gt_boxes', 'num_gt'} and bd['gt_boxes'].shape[2] == 5
    # aspect grouping: both images of a batch have the same orientation
    al.reset()
    for k in range(0, 6, 2):
        a, b = roidb[al.index[k]], roidb[al.index[k + 1]]
        assert (a['width'] >= a['height']) == (b['width'] >= b['height'])
    # precomputed proposals (FPN / alternate training)
    fcfg = C.experiment('rcnn_fpn_relation_learn_nms_8epoch')
    fcfg.SCALES[0] = (128, 192); fcfg.TRAIN.TOP_ROIS = 50
    rng = np.random.default_rng(4)
    props = [np.hstack((OB.clip_boxes(np.hstack((p, p + rng.uniform(10, 40, p.shape))), (r['height'], r['width'])), rng.uniform(0, 1, (30, 1))))
             for r in roidb for p in [rng.uniform(0, 100, (30, 2))]]
    rp = db.create_roidb_from_box_list(props, roidb, overlaps_fn=OB.bbox_overlaps)
    merged = coco_mod.merge_roidbs(rp, roidb)
    it = LD.ROIIter(merged, fcfg, batch_size=2)
    b = next(iter(it))
    assert b['proposals'].shape == (2, 50, 4) and b['gt_boxes'].shape[2] == 5 and int(b['num_gt'][0]) == 4
    # TOP_ROIS only truncates (core/rcnn.py:128-146 keeps the variable count): 30 real rows, zero padding, TRUE count reported
    assert b['num_proposals'].tolist() == [30, 30] and float(b['proposals'][:, 30:].abs().max()) == 0.0
    fcfg.TRAIN.TOP_ROIS = 20
    b20 = next(iter(LD.ROIIter(merged, fcfg, batch_size=2)))
    assert b20['proposals'].shape == (2, 20, 4) and b20['num_proposals'].tolist() == [20, 20]
    assert np.array_equal(b20['proposals'].numpy(), b['proposals'][:, :20].numpy())
    assert b['data'].shape[2] % 32 == 0 and b['data'].shape[3] % 32 == 0           # IMAGE_STRIDE 32
    sc = float(b['im_info'][0, 2])
    want = IMG.clip_boxes(np.round(merged[0]['boxes'][:30].astype(np.float64) * sc), (float(b['im_info'][0, 0]), float(b['im_info'][0, 1])))
    assert np.allclose(b['proposals'][0, :30].numpy(), want)


def _ev(gts, dts):
    ev = cocoeval.COCOeval(gts, dts)
    ev.evaluate(); ev.accumulate()
    return ev.summarize()


def test_cocoeval_known_answers():
    g = lambda i, c, b, crowd=0, iid=1: {'id': i, 'image_id': iid, 'category_id': c, 'bbox': b, 'area': b[2] * b[3], 'iscrowd': crowd}
    d = lambda c, b, s, iid=1: {'image_id': iid, 'category_id': c, 'bbox': b, 'score': s}
    gts = [g(1, 1, [10, 10, 50, 50]), g(2, 1, [100, 100, 40, 40]), g(3, 2, [30, 30, 100, 100])]
    perfect = [d(1, [10, 10, 50, 50], 0.9), d(1, [100, 100, 40, 40], 0.8), d(2, [30, 30, 100, 100], 0.7)]
    st = _ev(gts, perfect)
    assert np.allclose(st[:3], 1.0) and np.isclose(st[8], 1.0) and np.isclose(st[6], 0.75)       # AR@1: one of two persons, one car
    # IoU 0.6 true positive + an unmatched false positive below it: AP50 = 1, AP75 = 0, AP = 3 of 10 thresholds
    one = [g(1, 1, [0, 0, 100, 100])]
    st = _ev(one, [d(1, [0, 0, 100, 60], 0.9), d(1, [300, 300, 10, 10], 0.8)])
    assert np.isclose(st[1], 1.0) and np.isclose(st[2], 0.0) and np.isclose(st[0], 0.3)
    # the false positive ranks first: precision 0.5 at every recall point
    st = _ev(one, [d(1, [0, 0, 100, 100], 0.5), d(1, [300, 300, 10, 10], 0.8)])
    assert np.isclose(st[1], 0.5) and np.isclose(st[0], 0.5)
    # a detection inside a crowd region is ignored (neither TP nor FP); the regular gt is still found
    st = _ev([g(1, 1, [0, 0, 100, 100]), g(2, 1, [200, 200, 100, 100], crowd=1)], [d(1, [0, 0, 100, 100], 0.9), d(1, [210, 210, 30, 30], 0.95)])
    assert np.isclose(st[0], 1.0)
    # area ranges: a 20x20 object is 'small' only
    st = _ev([g(1, 1, [0, 0, 20, 20])], [d(1, [0, 0, 20, 20], 0.9)])
    assert np.isclose(st[3], 1.0) and st[4] == -1 and st[5] == -1
    # crowd IoU = intersection / detection area
    assert np.isclose(cocoeval.bbox_iou([[0, 0, 10, 10]], [[0, 0, 100, 100]], [1])[0, 0], 1.0)
    assert np.isclose(cocoeval.bbox_iou([[0, 0, 10, 10]], [[0, 0, 100, 100]], [0])[0, 0], 0.01)


def test_evaluate_detections_writes_results_and_scores(tmp_path):
    db = make_dataset(str(tmp_path), degenerate=False)
    all_boxes = [[np.zeros((0, 5), np.float32) for _ in range(db.num_images)] for _ in range(db.num_classes)]
    for i, iid in enumerate(db.image_set_index):        # detections = the annotations themselves, as xyxy boxes with +1 extents
        for a in db._anns[iid]:
            if a['iscrowd']:
                continue
            x, y, w, h = a['bbox']
            c = db._coco_ind_to_class_ind[a['category_id']]
            all_boxes[c][i] = np.vstack((all_boxes[c][i], np.array([[x, y, x + w - 1, y + h - 1, 0.9]], np.float32)))
    info, stats = db.evaluate_detections(all_boxes)
    res = json.load(open(os.path.join(db.result_path, 'results', 'detections_val2014_results.json')))
    assert len(res) == 18 and set(res[0]) == {'image_id', 'category_id', 'bbox', 'score'} and {r_['category_id'] for r_ in res} == {1, 3, 7}
    assert np.isclose(stats[0], 1.0, atol=1e-6) and np.isclose(stats[8], 1.0) and 'person' in info and 'AP50' in info


def test_cocoeval_matches_the_reference_evaluation_code():
    """dataset/cocoeval.py vs tests/golden/cocoeval.npz = the output of the reference's own lib/dataset/pycocotools/cocoeval.py
    (evaluate / evaluateImg / accumulate / summarize, executed by tests/golden/gen_golden.py with its C `mask.iou` replaced by
    a transcription of maskApi.c:bbIou) on a synthetic problem with crowd boxes, all area ranges, duplicates and misses:
    the 12 statistics AND the full precision [T,R,K,A,M] / recall [T,K,A,M] arrays agree to 1e-12."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    import cases
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'cocoeval.npz'))
    gts, dts = cases.cocoeval_case()
    ev = cocoeval.COCOeval(gts, dts)
    ev.evaluate(); ev.accumulate()
    st = ev.summarize()
    assert ev.eval['precision'].shape == g['precision'].shape and ev.eval['recall'].shape == g['recall'].shape
    np.testing.assert_allclose(ev.eval['recall'], g['recall'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(ev.eval['precision'], g['precision'], rtol=0, atol=1e-12)
    np.testing.assert_allclose(st, g['stats'], rtol=0, atol=1e-12)
    assert 0.1 < st[0] < 0.9 and st[1] > st[0] > st[2]                # a non-degenerate problem
