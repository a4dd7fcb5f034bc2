"""`learn_nms` operator -- device-resident replacement of relation_rcnn/operator_py/learn_nms.py
(same registered name, string attributes, 19/20-argument list, outputs and infer_shape,
learn_nms.py:404-458)."""
import numpy as np
import torch

from . import CustomOp, CustomOpProp, register
from ..learn_nms import LearnNMS

ARGS = ['cls_score', 'bbox_pred', 'rois', 'im_info', 'fc_all_2_relu', 'nms_rank_weight', 'nms_rank_bias',
        'roi_feat_embedding_weight', 'roi_feat_embedding_bias', 'nms_pair_pos_fc1_1_weight',
        'nms_pair_pos_fc1_1_bias', 'nms_query_1_weight', 'nms_query_1_bias', 'nms_key_1_weight',
        'nms_key_1_bias', 'nms_linear_out_1_weight', 'nms_linear_out_1_bias', 'nms_logit_weight', 'nms_logit_bias']


class LearnNmsOperator(CustomOp):
    def __init__(self, num_fg_classes, bbox_means, bbox_stds, first_n, class_agnostic, num_thresh, class_thresh,
                 nongt_dim=None, has_non_gt_index=False):
        super(LearnNmsOperator, self).__init__()
        assert class_agnostic, "class-specific regression is not used by any shipped cfg"
        self.num_fg_classes, self.first_n, self.num_thresh = num_fg_classes, first_n, num_thresh
        self.bbox_means, self.bbox_stds, self.class_thresh, self.nongt_dim = bbox_means, bbox_stds, class_thresh, nongt_dim
        self.has_non_gt_index = has_non_gt_index
        self._impl, self._key = None, None

    def forward(self, is_train, req, in_data, out_data, aux):
        cls_score, bbox_pred, rois, im_info, feat = in_data[:5]
        if self.nongt_dim is not None:                     # learn_nms.py:265-267, 282-283
            cls_score, bbox_pred, rois, feat = (t[:self.nongt_dim] for t in (cls_score, bbox_pred, rois, feat))
        elif self.has_non_gt_index:
            # learn_nms.py:268-270, 284-285: nd.take of the non-gt rows of cls_score / bbox_pred / rois (FPN training graphs).  As in the
            # reference (:335-339) the roi feature embedding is NOT gathered: the ranks index fc_all_2_relu's own rows, which is the
            # same thing whenever the non-gt rows are a prefix (they are in every graph the reference builds this operator into)
            idx = in_data[19].long()
            cls_score, bbox_pred, rois = (t.index_select(0, idx) for t in (cls_score, bbox_pred, rois))
        key = tuple(t.data_ptr() for t in in_data[5:19])
        if self._impl is None or key != self._key:
            params = dict(zip(ARGS[5:], in_data[5:19]))
            self._impl = LearnNMS(params, self.num_fg_classes, self.first_n, self.num_thresh, self.class_thresh,
                                  self.bbox_means, self.bbox_stds, dtype=feat.dtype if feat.dtype != torch.float64 else torch.float32,
                                  device=cls_score.device)
            self._key = key
        r = self._impl.forward(cls_score[None].float().contiguous(), bbox_pred[None].float().contiguous(), rois[None].float().contiguous(),
                               im_info.reshape(-1, 3).float().contiguous(), feat[None].contiguous(), want_detections=False)
        self.assign(out_data[0], req[0], r['nms_multi_score'][0])
        self.assign(out_data[1], req[1], r['sorted_bbox'][0])
        self.assign(out_data[2], req[2], r['sorted_score'][0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        for g, r in zip(in_grad, req):
            self.assign(g, r, 0)


@register('learn_nms')
class LearnNmsProp(CustomOpProp):
    def __init__(self, num_fg_classes, bbox_means, bbox_stds, first_n, class_agnostic, num_thresh, class_thresh,
                 nongt_dim, has_non_gt_index):
        super(LearnNmsProp, self).__init__(need_top_grad=False)
        self.num_fg_classes = int(num_fg_classes)
        self.nongt_dim = int(nongt_dim) if nongt_dim != 'None' else None
        self.class_thresh = float(class_thresh)
        assert ',' not in bbox_means and ',' not in bbox_stds
        if bbox_means == 'None' or bbox_stds == 'None':
            self.bbox_means = self.bbox_stds = None
        else:
            self.bbox_means = np.array([float(v) for v in bbox_means[1:-1].split()], dtype=float)
            self.bbox_stds = np.array([float(v) for v in bbox_stds[1:-1].split()], dtype=float)
        self.first_n = int(first_n)
        self.class_agnostic = class_agnostic == 'True'
        self.num_thresh = int(num_thresh)
        self.has_non_gt_index = has_non_gt_index == 'True'

    def list_arguments(self):
        return ARGS + (['non_gt_index'] if self.has_non_gt_index else [])

    def list_outputs(self):
        return ['nms_multi_score', 'sorted_bbox', 'sorted_score']

    def infer_shape(self, in_shape):
        return in_shape, [(self.first_n, self.num_fg_classes, self.num_thresh), (self.first_n, self.num_fg_classes, 4),
                          (self.first_n, self.num_fg_classes)]

    def create_operator(self, ctx, shapes, dtypes):
        return LearnNmsOperator(self.num_fg_classes, self.bbox_means, self.bbox_stds, self.first_n, self.class_agnostic,
                                self.num_thresh, self.class_thresh, self.nongt_dim, self.has_non_gt_index)

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return []
