#!/usr/bin/env python
"""Per-class soft-NMS + image top-100 (relnet_class_nms_topk + relnet_image_topk, the detector's post-processing) on the two kinds of
class posterior it can see: FLAT (what the benchmark's random-init heads produce: every roi is a candidate in every class) and PEAKED
(what a trained head produces: a roi scores in one or two classes, half of the rois are background).  The pruned form stops a class list
as soon as its next pick cannot reach the image's top 100 -- strongest on flat posteriors; this prints both so that the headline's
post-processing share can be read for trained weights too.
    python tools/nms_peaked.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: F401,E402
from relnet_amd import ops  # noqa: E402


def boxes_and_objects(B, N, G, gen):
    """N proposals per image scattered around G objects (jittered copies: what the RPN hands over), 600 x 1000 images."""
    cx = torch.rand(B, G, generator=gen) * 900 + 50
    cy = torch.rand(B, G, generator=gen) * 500 + 50
    w = torch.rand(B, G, generator=gen) * 300 + 40
    h = torch.rand(B, G, generator=gen) * 300 + 40
    obj = torch.randint(0, G, (B, N), generator=gen)
    j = lambda s: torch.randn(B, N, generator=gen) * s      # noqa: E731
    ocx, ocy = torch.gather(cx, 1, obj) + j(12), torch.gather(cy, 1, obj) + j(12)
    ow, oh = torch.gather(w, 1, obj) * torch.exp(j(0.15)), torch.gather(h, 1, obj) * torch.exp(j(0.15))
    b = torch.stack([(ocx - ow / 2).clamp(0, 999), (ocy - oh / 2).clamp(0, 599), (ocx + ow / 2).clamp(0, 999), (ocy + oh / 2).clamp(0, 599)], 2)
    return b.double(), obj


def posteriors(kind, B, N, C, G, obj, gen):
    if kind == 'flat':
        return torch.softmax(torch.randn(B, N, C, generator=gen) * 0.05, 2)
    cls_of_obj = torch.randint(1, C, (B, G), generator=gen)
    cls = torch.gather(cls_of_obj, 1, obj)
    bg = torch.rand(B, N, generator=gen) < 0.5
    conf = 1 - torch.exp(torch.rand(B, N, generator=gen) * -6.0) * 0.5          # 0.5 ... 0.9988 on the roi's own class (or background)
    logits = torch.randn(B, N, C, generator=gen) * 4.0                          # the rest on two or three confusable classes
    p = torch.softmax(logits, 2) * (1 - conf)[..., None]
    tgt = torch.where(bg, torch.zeros_like(cls), cls)
    p.scatter_add_(2, tgt[..., None], conf[..., None])
    return p / p.sum(2, keepdim=True)


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 108
    N, C, G = 300, 81, 8
    gen = torch.Generator().manual_seed(5)
    boxes, obj = boxes_and_objects(B, N, G, gen)
    boxes = boxes.cuda().contiguous()
    for kind in ('flat', 'peaked'):
        prob = posteriors(kind, B, N, C, G, obj, gen).float().cuda().contiguous()
        cand = int((prob[:, :, 1:] > 1e-3).sum().item())
        pruned = lambda: ops.image_topk(*ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100, top_k=100), 100)       # noqa: E731
        full = lambda: ops.image_topk(*ops.class_nms(prob, boxes, 1e-3, 0.6, True, max_picks=100), 100)                     # noqa: E731
        a, b_ = pruned()[0], full()[0]
        same = bool(torch.equal(a, b_))
        print('%-6s posterior, %d images: %.1f candidates (roi, class) per image above 1e-3; pruned class NMS + top-100 %.1f us, unpruned %.1f us '
              '(%.2f x), outputs identical: %s' % (kind, B, cand / B, timeit(pruned), timeit(full), timeit(full) / timeit(pruned), same))


if __name__ == '__main__':
    main()
