#!/usr/bin/env python
"""Class-NMS / image top-k probe: serial depth (picks per class) and time of relnet_class_nms_topk and relnet_image_topk on the
benchmark's flat random-init posteriors at 1 / 8 / 54 images.  python tools/nms_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: F401,E402
from relnet_amd import ops  # noqa: E402


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def case(B, N, C, seed, scale):
    g = torch.Generator().manual_seed(seed)
    x1 = torch.rand(B, N, generator=g) * 800; y1 = torch.rand(B, N, generator=g) * 450
    w = torch.rand(B, N, generator=g) * 300 + 10; h = torch.rand(B, N, generator=g) * 300 + 10
    boxes = torch.stack([x1, y1, x1 + w, y1 + h], 2).double()
    prob = torch.softmax(torch.randn(B, N, C, generator=g) * scale, 2).float()
    return prob.contiguous().cuda(), boxes.contiguous().cuda()


def main():
    for scale in (0.05, 1.0):
        for B in (1, 8, 54):
            prob, boxes = case(B, 300, 81, 5, scale)
            dets, counts = ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100, top_k=100)
            c = counts.float()
            t_p = timeit(lambda: ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100, top_k=100))
            t_f = timeit(lambda: ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100))
            t_k = timeit(lambda: ops.image_topk(dets, counts, 100))
            t_z = timeit(lambda: torch.zeros((B, 80, 300, 5), device='cuda', dtype=torch.float64))
            print('logit scale %.2f B %2d: picks/class max %d mean %.1f | pruned %.1f us, full(100) %.1f us, image_topk %.1f us, zero-fill %.1f us'
                  % (scale, B, int(c.max()), c.mean().item(), t_p, t_f, t_k, t_z), flush=True)


def detector_case():
    """The detector's own head outputs on random-init weights (what bench.py times)."""
    from relnet_amd import backbone, detector
    params = backbone.init_params(seed=1)
    det = detector.Detector(params, dtype=torch.bfloat16, device='cuda')
    g = torch.Generator().manual_seed(0)
    for B in (1, 4, 8, 16, 54):
        Bd = min(B, 6)
        data = torch.randn(Bd, 3, 600, 1000, generator=g).cuda()
        im_info = torch.tensor([[600, 1000, 1.0]] * Bd).cuda()
        out = det.forward(data, im_info)
        rep = (B + Bd - 1) // Bd          # more images than the forward pass ran: its outputs repeated (same depth distribution)
        prob = out['cls_prob'].repeat(rep, 1, 1)[:B].contiguous()
        boxes = out['pred_boxes'].repeat(rep, 1, 1)[:B].contiguous()
        c = out['class_counts'].float()
        pr = prob[..., 1:]
        uniq = [int(torch.unique(pr[b]).numel()) for b in range(B)]
        t_p = timeit(lambda: ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100, top_k=100))
        dets, counts = ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100, top_k=100)
        t_k = timeit(lambda: ops.image_topk(dets, counts, 100))
        print('detector B %d: picks/class max %d mean %.1f total %d | prob min %.5f max %.5f distinct values %s | pruned %.1f us image_topk %.1f us thresh %s'
              % (B, int(c.max()), c.mean().item(), int(c.sum()), pr.min().item(), pr.max().item(), uniq, t_p, t_k,
                 out['image_thresh'].tolist()), flush=True)


if __name__ == '__main__':
    main()
    detector_case()
