"""Test-time drivers of `relation_rcnn/core/tester.py` on the HIP detector.

  pred_eval            :163-307  loop over a TestLoader, collect all_boxes[cls][image] = [k,5], `imdb.evaluate_detections`.
                                 The per-class threshold / (soft-)NMS / max_per_image steps (:244-277) run in the detector's
                                 post-processing kernels (relnet_class_nms / relnet_image_topk); learn-NMS outputs are
                                 thresholded as :231-242.
  generate_proposals   :63-126   RPN-only pass, boxes mapped back to the original image scale, written as the
                                 `<name>_rpn.pkl` list the precomputed-proposal training reads (imdb.load_rpn_data).
"""
import os
import pickle
import time

import numpy as np
import torch


def _class_lists(out, b, num_classes, scale):
    """detections [n,6] (class, score, x1,y1,x2,y2) of image b -> list over classes of [k,5] (x1,y1,x2,y2,score)."""
    n = int(out['num_detections'][b])
    det = out['detections'][b, :n].detach().float().cpu().numpy()
    res = [np.zeros((0, 5), np.float32) for _ in range(num_classes)]
    for c in range(1, num_classes):
        rows = det[det[:, 0] == c]
        if len(rows):
            res[c] = np.hstack((rows[:, 2:6], rows[:, 1:2])).astype(np.float32)
    return res


def pred_eval(detector, test_data, imdb, vis=False, thresh=1e-3, logger=None, device='cuda'):
    """Returns (info_str, stats, all_boxes).  `detector`: relnet_amd.detector.Detector (HAS_RPN graphs) or FPNDetector
    (precomputed proposals from the loader).  The detector divides boxes by im_info[2] itself (tester.py:156)."""
    num_images = imdb.num_images
    all_boxes = [[[] for _ in range(num_images)] for _ in range(imdb.num_classes)]
    t_net = []
    with torch.no_grad():
        for batch in test_data:
            t0 = time.time()
            data, im_info = batch['data'].to(device), batch['im_info'].to(device)
            if 'proposals' in batch:
                nprop = batch.get('num_proposals')                       # TestLoader pads the images' proposal lists to one length
                out = detector.forward(data, batch['proposals'].to(device), im_info,
                                       num_proposals=None if nprop is None else nprop.to(device=device, dtype=torch.int32))
            else:
                detector.im_hw = (int(data.shape[2]), int(data.shape[3]))
                out = detector.forward(data, im_info)
            torch.cuda.synchronize()
            t_net.append(time.time() - t0)
            for b, idx in enumerate(batch['index']):
                per_cls = _class_lists(out, b, imdb.num_classes, float(batch['im_info'][b, 2]))
                for c in range(1, imdb.num_classes):
                    all_boxes[c][idx] = per_cls[c]
    for c in range(imdb.num_classes):
        for i in range(num_images):
            if len(all_boxes[c][i]) == 0:
                all_boxes[c][i] = np.zeros((0, 5), np.float32)
    det_file = os.path.join(imdb.result_path, imdb.name + '_detections.pkl')
    with open(det_file, 'wb') as f:
        pickle.dump(all_boxes, f, protocol=pickle.HIGHEST_PROTOCOL)
    info, stats = imdb.evaluate_detections(all_boxes)
    if logger:
        logger.info('evaluate detections: \n{}'.format(info))
        logger.info('net time per batch: %.4f s' % (float(np.mean(t_net)) if t_net else 0.0))
    return info, stats, all_boxes


def generate_proposals(detector, test_data, imdb, thresh=0.0, device='cuda', save=True):
    """tester.py:63-126: per image [n,5] = (x1,y1,x2,y2 at the ORIGINAL scale, rpn score); returns the list and writes
    `<rpn_path>/rpn_data/<name>_rpn.pkl` (+ `_full_rpn.pkl` when thresh > 0)."""
    from ..operator_py.proposal import propose_batch
    imdb_boxes, original = [None] * imdb.num_images, [None] * imdb.num_images
    c = detector.cfg
    with torch.no_grad():
        for batch in test_data:
            data, im_info = batch['data'].to(device), batch['im_info'].to(device)
            f = detector.backbone.forward(data)
            rois, scores = propose_batch(f['rpn_cls_score'].float(), f['rpn_bbox_pred'].float(), im_info, detector.anchors,
                                         c.feat_stride, c.rpn_pre_nms_top_n, c.rpn_post_nms_top_n, c.rpn_nms_thresh, c.rpn_min_size,
                                         im_hw=(int(data.shape[2]), int(data.shape[3])), softmax_pairs=True)
            for b, idx in enumerate(batch['index']):
                boxes = rois[b, :, 1:].cpu().numpy() / float(batch['im_info'][b, 2])
                dets = np.hstack((boxes, scores[b].cpu().numpy().reshape(-1, 1))).astype(np.float32)
                original[idx] = dets
                imdb_boxes[idx] = dets[np.where(dets[:, 4] > thresh)[0], :]
    assert all(x is not None for x in imdb_boxes), 'calculations not complete'
    if save:
        imdb.save_rpn_data(imdb_boxes)
        if thresh > 0:
            imdb.save_rpn_data(original, full=True)
    return imdb_boxes
