"""Data side on the GPU: `pred_eval` (loader -> HIP detector -> COCO bbox evaluation) and the
precomputed-proposal flow `generate_proposals` -> `<name>_rpn.pkl` -> `rpn_roidb(append_gt)` -> `ROIIter` -> FPN training
step, on a miniature synthetic COCO tree (random-init weights: the numbers are meaningless, the plumbing is what is tested)."""
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_dataset import make_dataset  # noqa: E402

pytestmark = pytest.mark.gpu


def test_pred_eval_and_precomputed_proposal_training(tmp_path):
    import relnet_amd  # noqa: F401
    from relnet_amd import backbone, detector, train, config as C
    from relnet_amd.dataset import loader as LD, tester as TS
    db = make_dataset(str(tmp_path), degenerate=False)
    roidb = db.gt_roidb()
    cfg = C.experiment('rcnn_end2end_relation_8epoch')
    cfg.SCALES[0] = (128, 192)
    p = backbone.init_params(seed=2, num_classes=db.num_classes)
    g = torch.Generator().manual_seed(3)
    for k in ('cls_score_weight', 'bbox_pred_weight'):
        p[k] = torch.randn(p[k].shape, generator=g) * 0.05
    dcfg = detector.Config()
    dcfg.num_classes, dcfg.rpn_post_nms_top_n, dcfg.rpn_min_size = db.num_classes, 60, 8      # (random-init RPN: drop degenerate boxes)
    det = detector.Detector(p, dtype=torch.bfloat16, cfg=dcfg, im_hw=(128, 192))
    # ---- test.py / pred_eval: every image through the detector, all_boxes[cls][image], results json, 12 COCO stats
    info, stats, all_boxes = TS.pred_eval(det, LD.TestLoader(roidb, cfg, batch_size=1, has_rpn=True), db)
    assert len(all_boxes) == db.num_classes and len(all_boxes[1]) == db.num_images
    n_det = sum(len(all_boxes[c][i]) for c in range(1, db.num_classes) for i in range(db.num_images))
    assert 0 < n_det <= 100 * db.num_images and stats.shape == (12,) and (stats >= -1).all() and (stats <= 1).all()
    assert os.path.exists(os.path.join(db.result_path, 'results', 'detections_val2014_results.json'))
    for c in range(1, db.num_classes):                  # boxes are in original-image coordinates (tester.py:156)
        for i in range(db.num_images):
            b = all_boxes[c][i]
            assert b.shape[1] == 5
            if len(b):                   # (a decoded width < 1 px gives x2 = x1 - (1 - w) / im_scale, as in the reference's bbox_pred)
                assert (b[:, 2] >= b[:, 0] - 2.0).all()
                assert b[:, 2].max() <= roidb[i]['width'] + 0.5 and b[:, 3].max() <= roidb[i]['height'] + 0.5    # (resized extent - 1) / scale
    # ---- generate_proposals: RPN pass -> <name>_rpn.pkl in the reference's format
    boxes = TS.generate_proposals(det, LD.TestLoader(roidb, cfg, batch_size=2, has_rpn=True), db)
    assert len(boxes) == db.num_images and boxes[0].shape == (60, 5) and os.path.exists(db.rpn_file())
    assert boxes[0][:, 2].max() <= roidb[0]['width'] + 0.5 and boxes[0][0, 4] == boxes[0][:, 4].max()     # best RPN score first (short lists are padded cyclically)
    # ---- alternate / FPN training on the stored proposals: rpn_roidb(append_gt) -> ROIIter -> FPNTrainer step
    fcfg = C.experiment('rcnn_fpn_relation_learn_nms_8epoch')
    fcfg.SCALES[0] = (128, 192); fcfg.TRAIN.TOP_ROIS = 48
    merged = db.rpn_roidb(db.gt_roidb(), append_gt=True, top_roi=48)            # bbox_overlaps on the device twin
    assert merged[0]['boxes'].shape[0] == 48 + len(roidb[0]['boxes']) and merged[0]['is_gt'].sum() == len(roidb[0]['boxes'])
    pf = backbone.init_params(seed=4, fpn=True, num_classes=db.num_classes)
    tcfg = train.TrainConfig(); tcfg.learn_nms, tcfg.first_n, tcfg.num_classes = True, 16, db.num_classes
    tr = train.FPNTrainer(pf, tcfg)
    it = LD.ROIIter(merged, fcfg, batch_size=2, shuffle=True, aspect_grouping=True, seed=1, device='cuda')
    batch = next(iter(it))
    out = tr.forward_backward(batch['data'], batch['im_info'], batch['gt_boxes'], batch['proposals'], num_gt=batch['num_gt'],
                              num_proposals=batch['num_proposals'])
    assert out['rois'].shape[1] == 48 + batch['gt_boxes'].shape[1] and torch.isfinite(out['bbox_loss']).all()
    # short proposal lists: the padded rows sit behind the real ones, carry a zero box and never a label
    for b_ in range(2):
        n = int(batch['num_proposals'][b_])
        assert (out['rois'][b_, n:48, 1:] == 0).all() and (out['label'][b_, n:48] == -1).all()
    bad = [n for n in tr.W.slices if not torch.isfinite(tr.W.view(tr.W.grad, n)).all()]
    assert not bad, ('non-finite gradients', bad[:6], {k: float(v) for k, v in out.items() if k.endswith('loss')})
    tr.all_reduce(); tr.update()
    assert torch.isfinite(tr.W.master).all()
