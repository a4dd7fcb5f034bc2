"""CPU restatement of the training losses (TEST INFRASTRUCTURE ONLY).

Forward values follow the reference graph (symbols/resnet_v1_101_rcnn_attention_1024_pairwise_position_multi_head_16_
learn_nms.py:268-278, 364-392, 536-551); gradients are obtained by torch-CPU autograd of the scalar each MXNet loss
head implicitly minimises, so they do not share hand-derived formulas with the HIP kernels:
  SoftmaxOutput(normalization='valid', use_ignore)  -> sum of cross entropies over non-ignored rows / #valid * grad_scale
  MakeLoss(x, grad_scale)                           -> grad_scale * sum(x)
MXNet built-ins (SoftmaxOutput, smooth_l1, MakeLoss) are un-vendored: v1.1.0 semantics, PARITY UNPINNED.
"""
import numpy as np
import torch


def softmax_output(data, label, multi_output=False, use_ignore=False, ignore_label=-1.0, grad_scale=1.0):
    x = torch.tensor(np.asarray(data, np.float32), requires_grad=True)
    lab = torch.as_tensor(np.asarray(label, np.float32))
    if multi_output:
        B, C = x.shape[:2]
        logits = x.reshape(B, C, -1).permute(0, 2, 1).reshape(-1, C)
    else:
        logits = x.reshape(x.shape[0], -1)
    lab = lab.reshape(-1)
    valid = (lab != ignore_label) if use_ignore else torch.ones_like(lab, dtype=torch.bool)
    prob = torch.softmax(x, dim=1)
    logp = torch.log_softmax(logits.double(), dim=1)
    idx = lab.clamp(min=0).long()
    ce = -(logp[torch.arange(len(lab)), idx] * valid.double()).sum()
    (ce * grad_scale / max(int(valid.sum()), 1)).backward()
    return prob.detach().numpy(), x.grad.numpy()


def smooth_l1(x, sigma):
    s2 = sigma * sigma
    return torch.where(x.abs() < 1.0 / s2, 0.5 * s2 * x * x, x.abs() - 0.5 / s2)


def smooth_l1_loss(pred, target, weight, sigma, grad_scale):
    p = torch.tensor(np.asarray(pred, np.float32), requires_grad=True)
    w = torch.as_tensor(np.broadcast_to(np.asarray(weight, np.float32), p.shape).copy())
    loss = w * smooth_l1(p - torch.as_tensor(np.asarray(target, np.float32)), sigma)
    (loss.double().sum() * grad_scale).backward()
    return loss.detach().numpy(), p.grad.numpy()


def nms_loss(score, target, first_n, num_thresh, nms_loss_scale=1.0, nms_pos_scale=4.0, eps=1e-8):
    s = torch.tensor(np.asarray(score, np.float32), requires_grad=True)
    t = torch.as_tensor(np.asarray(target, np.float32))
    k = nms_loss_scale / float(first_n * num_thresh)
    pos = k * (-(t * torch.log(s + eps)))
    neg = k * (-((1.0 - t) * torch.log(1.0 - s + eps)))
    (nms_pos_scale * pos.double().sum() + neg.double().sum()).backward()
    return pos.detach().numpy(), neg.detach().numpy(), s.grad.numpy()
