"""`proposal_target`, `BoxAnnotatorOHEM`, `nms_multi_target` -- device-resident replacements of
relation_rcnn/operator_py/{proposal_target,box_annotator_ohem,nms_multi_target}.py with the
reference's registered names, string attributes, argument / output lists and shapes."""
import numpy as np
import torch

from . import CustomOp, CustomOpProp, register
from .. import ops


class ProposalTargetOperator(CustomOp):
    def __init__(self, num_classes, batch_images, batch_rois, cfg, fg_fraction):
        super(ProposalTargetOperator, self).__init__()
        self._num_classes, self._batch_images, self._batch_rois, self._cfg = num_classes, batch_images, batch_rois, cfg
        if batch_rois != -1:
            raise NotImplementedError("only BATCH_ROIS = -1 (all proposals + gt, no sampling) is on the hot path; "
                                      "every shipped end2end cfg uses it")

    def forward(self, is_train, req, in_data, out_data, aux):
        rois, gt = in_data[0].float(), in_data[1].float()
        if not bool((rois[:, 0] == 0).all()):
            raise AssertionError('Only single item batches are supported')      # proposal_target.py:69
        c = self._cfg
        r, lab, bt, bw = ops.proposal_target(rois[None], gt[None], None, self._num_classes, c['class_agnostic'],
                                             c['bg_thresh_hi'], c['means'], c['stds'], c['weights'])
        for ind, val in enumerate([r[0], lab[0], bt[0], bw[0]]):
            self.assign(out_data[ind], req[ind], val)

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], 0)


@register('proposal_target')
class ProposalTargetProp(CustomOpProp):
    """cfg: the reference pickles its whole EasyDict into this attribute (proposal_target.py:107); here a
    plain dict (or its repr) with class_agnostic / bg_thresh_hi / means / stds / weights."""

    def __init__(self, num_classes, batch_images, batch_rois, cfg='None', fg_fraction='0.25'):
        super(ProposalTargetProp, self).__init__(need_top_grad=False)
        self._num_classes, self._batch_images, self._batch_rois = int(num_classes), int(batch_images), int(batch_rois)
        d = dict(class_agnostic=True, bg_thresh_hi=0.5, means=(0., 0., 0., 0.), stds=(0.1, 0.1, 0.2, 0.2), weights=(1., 1., 1., 1.))
        if cfg not in (None, 'None'):
            if isinstance(cfg, str):
                import ast
                cfg = ast.literal_eval(cfg)
            if isinstance(cfg, bytes):                        # the reference's form: cPickle.dumps(config tree), SYM_REL:220
                import pickle
                cfg = pickle.loads(cfg)
            if 'TRAIN' in cfg:                                # reference config tree (config/config.py) -> the fields used
                t = cfg['TRAIN']                              # by sample_rois_v2 / expand_bbox_regression_targets
                norm = bool(t['BBOX_NORMALIZATION_PRECOMPUTED'])
                cfg = dict(class_agnostic=bool(cfg['CLASS_AGNOSTIC']), bg_thresh_hi=float(t['BG_THRESH_HI']),
                           means=tuple(float(v) for v in t['BBOX_MEANS']) if norm else (0., 0., 0., 0.),
                           stds=tuple(float(v) for v in t['BBOX_STDS']) if norm else (1., 1., 1., 1.),
                           weights=tuple(float(v) for v in t['BBOX_WEIGHTS']))
            d.update(cfg)
        self._cfg, self._fg_fraction = d, float(fg_fraction)

    def list_arguments(self):
        return ['rois', 'gt_boxes']

    def list_outputs(self):
        return ['rois_output', 'label', 'bbox_target', 'bbox_weight']

    def infer_shape(self, in_shape):
        rpn_rois_shape, gt_boxes_shape = in_shape[0], in_shape[1]
        rois = rpn_rois_shape[0] + gt_boxes_shape[0]                     # BATCH_ROIS = -1 (proposal_target.py:117-118)
        return [rpn_rois_shape, gt_boxes_shape], [(rois, 5), (rois,), (rois, self._num_classes * 4), (rois, self._num_classes * 4)]

    def create_operator(self, ctx, shapes, dtypes):
        return ProposalTargetOperator(self._num_classes, self._batch_images, self._batch_rois, self._cfg, self._fg_fraction)


class BoxAnnotatorOHEMOperator(CustomOp):
    def __init__(self, num_classes, num_reg_classes, roi_per_img):
        super(BoxAnnotatorOHEMOperator, self).__init__()
        self._num_classes, self._num_reg_classes, self._roi_per_img = num_classes, num_reg_classes, roi_per_img

    def forward(self, is_train, req, in_data, out_data, aux):
        lo, wo = ops.box_annotator_ohem(*[t.float()[None] for t in in_data[:5]], roi_per_img=self._roi_per_img)
        self.assign(out_data[0], req[0], lo[0])
        self.assign(out_data[1], req[1], wo[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(len(in_grad)):
            self.assign(in_grad[i], req[i], 0)


@register('BoxAnnotatorOHEM')
class BoxAnnotatorOHEMProp(CustomOpProp):
    def __init__(self, num_classes, num_reg_classes, roi_per_img):
        super(BoxAnnotatorOHEMProp, self).__init__(need_top_grad=False)
        self._num_classes, self._num_reg_classes, self._roi_per_img = int(num_classes), int(num_reg_classes), int(roi_per_img)

    def list_arguments(self):
        return ['cls_score', 'bbox_pred', 'labels', 'bbox_targets', 'bbox_weights']

    def list_outputs(self):
        return ['labels_ohem', 'bbox_weights_ohem']

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[2], in_shape[4]]

    def create_operator(self, ctx, shapes, dtypes):
        return BoxAnnotatorOHEMOperator(self._num_classes, self._num_reg_classes, self._roi_per_img)


class NmsMultiTargetOp(CustomOp):
    def __init__(self, target_thresh):
        super(NmsMultiTargetOp, self).__init__()
        self._target_thresh = target_thresh

    def forward(self, is_train, req, in_data, out_data, aux):
        bbox, gt_box, score = in_data[0].float(), in_data[1].float(), in_data[2].float()
        batch_image, num_gt, code_size = gt_box.shape
        assert batch_image == 1, 'only support batch_image=1, but receive %d' % num_gt
        assert code_size == 5, 'code_size of gt should be 5, but receive %d' % code_size
        assert score.dim() == 2 and score.shape[1] == bbox.shape[1]
        out = ops.nms_multi_target(bbox[None], gt_box, score[None], None, tuple(self._target_thresh))
        self.assign(out_data[0], req[0], out[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for i in range(3):
            self.assign(in_grad[i], req[i], 0)


@register('nms_multi_target')
class NmsMultiTargetProp(CustomOpProp):
    def __init__(self, target_thresh):
        super(NmsMultiTargetProp, self).__init__(need_top_grad=False)
        self._target_thresh = np.array([float(v) for v in target_thresh[1:-1].replace(',', ' ').split()], dtype=float)

    def list_arguments(self):
        return ['bbox', 'gt_bbox', 'score']

    def list_outputs(self):
        return ['nms_multi_target']

    def infer_shape(self, in_shape):
        bbox_shape, score_shape = in_shape[0], in_shape[2]
        assert bbox_shape[0] == score_shape[0], 'ROI number should be same for bbox and score'
        return in_shape, [(bbox_shape[0], bbox_shape[1], len(self._target_thresh))]

    def create_operator(self, ctx, shapes, dtypes):
        return NmsMultiTargetOp(self._target_thresh)
