"""Training losses of the reference graphs on device tensors: forward value + the gradient with respect to the
network output (what MXNet's backward pass starts from).  Kernels: csrc/losses.hip.

  softmax_output   mx.sym.SoftmaxOutput(data, label, normalization='valid', use_ignore, ignore_label, multi_output,
                   grad_scale)                                  -- rpn_cls_prob, cls_prob
  smooth_l1_loss   mx.sym.MakeLoss(weight * mx.sym.smooth_l1(scalar, data=pred - target), grad_scale)
                                                                -- rpn_bbox_loss, bbox_loss
  nms_loss         nms_pos_loss / nms_neg_loss of the learn-NMS head
(symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16_learn_nms.py:268-278, 364-392, 536-551)
"""
import torch

from . import lib as _lib
from .ops import _chk, _ptr, _stream


def softmax_output(data, label=None, multi_output=False, use_ignore=False, ignore_label=-1.0, grad_scale=1.0,
                   normalization='valid', want_grad=True, group=None):
    """data [n, C] (or [B, C, ...] with multi_output: softmax over axis 1); label [n] / [B, ...] float class ids.
    Returns (prob, grad) with grad = d(sum of cross entropies)/d(data) normalised like MXNet ('valid').
    group: positions per normalisation group (default: the whole call).  The reference runs one image per executor, so its
    'valid' count is per image: a batched call passes the positions of one image (rois per image, or A*h*w anchors)."""
    _chk(data, label)
    if normalization != 'valid':
        raise NotImplementedError("only normalization='valid' (the reference's setting) is built")
    assert data.dtype == torch.float32 and data.is_contiguous()
    if multi_output:
        outer, Cn = data.shape[0], data.shape[1]
        inner = data[0, 0].numel()
    else:
        outer, Cn, inner = data.shape[0], data[0].numel(), 1
    want_grad = want_grad and label is not None
    if label is not None:
        label = label.to(torch.float32).contiguous()
        if label.numel() != outer * inner:
            raise ValueError("label has %d entries, expected %d" % (label.numel(), outer * inner))
    prob = torch.empty_like(data)
    grad = torch.empty_like(data) if want_grad else None
    total = outer * inner
    group = total if not group else int(group)
    cnt = torch.empty(total // group, device=data.device, dtype=torch.int32) if want_grad else None
    _lib.call('relnet_softmax_output_ex', data.data_ptr(), _ptr(label), prob.data_ptr(), _ptr(grad), _ptr(cnt), outer, Cn,
              inner, int(use_ignore), float(ignore_label), float(grad_scale), group, _stream())
    return prob, grad


def smooth_l1_loss(pred, target, weight=None, scalar=1.0, grad_scale=1.0):
    """-> (loss tensor = weight * smooth_l1(pred - target), grad wrt pred = grad_scale * weight * smooth_l1')."""
    _chk(pred, target, weight)
    assert pred.dtype == torch.float32 and pred.is_contiguous() and target.shape == pred.shape
    target = target.to(torch.float32).contiguous()
    if weight is not None:
        weight = weight.to(torch.float32).expand_as(pred).contiguous()
    loss, grad = torch.empty_like(pred), torch.empty_like(pred)
    _lib.call('relnet_smooth_l1_loss', pred.data_ptr(), target.data_ptr(), _ptr(weight), loss.data_ptr(), grad.data_ptr(),
              pred.numel(), float(scalar), float(grad_scale), _stream())
    return loss, grad


def nms_loss(nms_multi_score, nms_multi_target, first_n, num_thresh, nms_loss_scale=1.0, nms_pos_scale=4.0, eps=1e-8):
    """-> (nms_pos_loss, nms_neg_loss, grad wrt nms_multi_score)."""
    _chk(nms_multi_score, nms_multi_target)
    s = nms_multi_score.to(torch.float32).contiguous()
    t = nms_multi_target.to(torch.float32).contiguous()
    assert s.shape == t.shape
    pos, neg, grad = torch.empty_like(s), torch.empty_like(s), torch.empty_like(s)
    _lib.call('relnet_nms_loss', s.data_ptr(), t.data_ptr(), pos.data_ptr(), neg.data_ptr(), grad.data_ptr(), s.numel(),
              float(eps), float(nms_loss_scale) / float(first_n * num_thresh), float(nms_pos_scale), _stream())
    return pos, neg, grad
