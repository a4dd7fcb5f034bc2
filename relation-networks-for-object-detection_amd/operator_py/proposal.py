"""`proposal` operator -- device-resident replacement of relation_rcnn/operator_py/proposal.py.

Same registered name, constructor attributes (all strings), list_arguments/outputs,
infer_shape and error behaviour (ValueError for batch > 1 when used through the reference's
single-image protocol, proposal.py:54-56); forward() runs decode -> top-K sort -> bitmask NMS
-> greedy scan entirely on the GPU (the reference does four device<->host round trips and a
numpy scan).  `propose_batch` is the batched (B images per launch) entry the detector uses.
"""
import numpy as np
import torch

from . import CustomOp, CustomOpProp, register
from .. import ops


def generate_anchors(base_size=16, ratios=(0.5, 1, 2), scales=(8, 16, 32)):
    """Base anchors, ratio-major / scale-minor, around the (0,0,base-1,base-1) window
    (lib/rpn/generate_anchor.py:22-86).  Host-side constant table, float64."""
    ratios = np.asarray(ratios, dtype=np.float64)
    scales = np.asarray(scales, dtype=np.float64)

    def whctrs(a):
        w, h = a[2] - a[0] + 1.0, a[3] - a[1] + 1.0
        return w, h, a[0] + 0.5 * (w - 1.0), a[1] + 0.5 * (h - 1.0)

    def mk(ws, hs, cx, cy):
        ws, hs = np.reshape(ws, (-1, 1)), np.reshape(hs, (-1, 1))
        return np.hstack((cx - 0.5 * (ws - 1), cy - 0.5 * (hs - 1), cx + 0.5 * (ws - 1), cy + 0.5 * (hs - 1)))

    w, h, cx, cy = whctrs(np.array([0, 0, base_size - 1, base_size - 1], dtype=np.float64))
    ws = np.round(np.sqrt(w * h / ratios))
    hs = np.round(ws * ratios)
    out = []
    for ra in mk(ws, hs, cx, cy):
        w, h, cx, cy = whctrs(ra)
        out.append(mk(w * scales, h * scales, cx, cy))
    return np.vstack(out)


def _parse_tuple(s):
    s = s.strip()
    return np.array([float(x) for x in s[1:-1].replace(' ', '').split(',') if x], dtype=np.float64)


def propose_batch(cls_prob, bbox_pred, im_info, anchors, feat_stride, pre_nms_top_n, post_nms_top_n,
                  threshold, min_size, want_debug=False, im_hw=None, softmax_pairs=False, want_num=False):
    """Batched proposal: cls_prob [B,2A,H,W], bbox_pred [B,4A,H,W], im_info [B,3] ->
    rois [B, post, 5] (column 0 = image index in the batch), scores [B, post] (want_num: + num_keep [B] int32, the
    number of boxes that survived NMS before padding: where it is < post the reference pads with `npr.choice` (random,
    proposal.py:154-156) and this operator with keep[i mod num_keep] -- a documented, unavoidable difference)."""
    boxes, scores = ops.proposal_decode(cls_prob, bbox_pred, im_info, anchors, feat_stride, min_size,
                                        im_hw=im_hw, softmax_pairs=softmax_pairs)
    n = scores.shape[1]
    k = min(pre_nms_top_n, n) if pre_nms_top_n > 0 else n
    det, index, count = ops.topk_sort(scores, boxes, k)
    if 0 < post_nms_top_n <= 2048:      # fused greedy NMS: no n x n bitmask, stops at post_nms_top_n keeps
        r = ops.nms_greedy(det, threshold, post_nms_top_n, counts=count, want_keep=want_debug)
    else:
        r = ops.nms_sorted(det, threshold, post=post_nms_top_n, counts=count, want_keep=want_debug)
    if want_debug:
        return r['rois'], r['scores'], dict(det=det, order=index, keep=r['keep'], num_keep=r['num_keep'])
    if want_num:
        return r['rois'], r['scores'], r['num_keep']
    return r['rois'], r['scores']


class ProposalOperator(CustomOp):
    def __init__(self, feat_stride, scales, ratios, output_score, rpn_pre_nms_top_n,
                 rpn_post_nms_top_n, threshold, rpn_min_size):
        super(ProposalOperator, self).__init__()
        self._feat_stride = feat_stride
        self._scales = _parse_tuple(scales)
        self._ratios = _parse_tuple(ratios)
        self._anchors = generate_anchors(base_size=self._feat_stride, scales=self._scales, ratios=self._ratios)
        self._num_anchors = self._anchors.shape[0]
        self._output_score = output_score
        self._rpn_pre_nms_top_n = rpn_pre_nms_top_n
        self._rpn_post_nms_top_n = rpn_post_nms_top_n
        self._threshold = threshold
        self._rpn_min_size = rpn_min_size
        self._anchors_dev = None

    def forward(self, is_train, req, in_data, out_data, aux):
        batch_size = in_data[0].shape[0]
        if batch_size > 1:
            raise ValueError("Sorry, multiple images each device is not implemented")
        dev = in_data[0].device
        if self._anchors_dev is None or self._anchors_dev.device != dev:
            self._anchors_dev = torch.as_tensor(self._anchors, dtype=torch.float64, device=dev)
        rois, scores = propose_batch(in_data[0].float(), in_data[1].float(), in_data[2].float().reshape(-1, 3),
                                     self._anchors_dev, self._feat_stride, self._rpn_pre_nms_top_n,
                                     self._rpn_post_nms_top_n, self._threshold, self._rpn_min_size)
        self.assign(out_data[0], req[0], rois[0])
        if self._output_score:
            self.assign(out_data[1], req[1], scores[0].reshape(-1, 1))

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        self.assign(in_grad[0], req[0], 0)
        self.assign(in_grad[1], req[1], 0)
        self.assign(in_grad[2], req[2], 0)


@register("proposal")
class ProposalProp(CustomOpProp):
    def __init__(self, feat_stride='16', scales='(8, 16, 32)', ratios='(0.5, 1, 2)', output_score='False',
                 rpn_pre_nms_top_n='6000', rpn_post_nms_top_n='300', threshold='0.3', rpn_min_size='16'):
        super(ProposalProp, self).__init__(need_top_grad=False)
        self._feat_stride = int(feat_stride)
        self._scales = scales
        self._ratios = ratios
        self._output_score = str(output_score) in ('True', 'true', '1')
        self._rpn_pre_nms_top_n = int(rpn_pre_nms_top_n)
        self._rpn_post_nms_top_n = int(rpn_post_nms_top_n)
        self._threshold = float(threshold)
        self._rpn_min_size = int(rpn_min_size)

    def list_arguments(self):
        return ['cls_prob', 'bbox_pred', 'im_info']

    def list_outputs(self):
        return ['output', 'score'] if self._output_score else ['output']

    def infer_shape(self, in_shape):
        cls_prob_shape, bbox_pred_shape = in_shape[0], in_shape[1]
        assert cls_prob_shape[0] == bbox_pred_shape[0], 'ROI number does not equal in cls and reg'
        batch_size = cls_prob_shape[0]
        im_info_shape = (batch_size, 3)
        output_shape = (self._rpn_post_nms_top_n, 5)
        score_shape = (self._rpn_post_nms_top_n, 1)
        if self._output_score:
            return [cls_prob_shape, bbox_pred_shape, im_info_shape], [output_shape, score_shape]
        return [cls_prob_shape, bbox_pred_shape, im_info_shape], [output_shape]

    def create_operator(self, ctx, shapes, dtypes):
        return ProposalOperator(self._feat_stride, self._scales, self._ratios, self._output_score,
                                self._rpn_pre_nms_top_n, self._rpn_post_nms_top_n, self._threshold,
                                self._rpn_min_size)

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return []
