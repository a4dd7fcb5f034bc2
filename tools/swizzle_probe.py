#!/usr/bin/env python
"""Workgroup order of the wide convolution layers (N >= 512: several 256-column tiles per row panel) at 54 images: the XCD-aware
order (all column tiles of a row panel on ONE XCD: the activations are shared through its L2) against the plain order (column tile
c on XCD c % 8: a filter tile stays in that L2 while the row panels stream past).  python tools/swizzle_probe.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: F401,E402
from relnet_amd import ops, lib  # noqa: E402

L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 54


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def case(H, W, Cin, Cout, k, dil, resid=False):
    x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.03).to(torch.bfloat16)
    b = torch.randn(Cout, device='cuda')
    out = torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16)
    return lambda: ops.conv2d_nhwc(x, w, b, ksize=k, pad=dil if k == 3 else 0, dil=dil, relu=True, out=out)


def main():
    for name, args in (('res5a projection 1024->2048', (38, 63, 1024, 2048, 1, 1)), ('res5 3x3 512 d2', (38, 63, 512, 512, 3, 2)),
                       ('rpn 3x3 1024->512', (38, 63, 1024, 512, 3, 1)), ('res5 reduce 2048->512', (38, 63, 2048, 512, 1, 1)),
                       ('res5a reduce 1024->512', (38, 63, 1024, 512, 1, 1)), ('res4a projection 512->1024', (38, 63, 512, 1024, 1, 1)),
                       ('res4 3x3 256', (38, 63, 256, 256, 3, 1)), ('res3a projection 256->512 (75x125)', (75, 125, 256, 512, 1, 1))):
        fn = case(*args)
        row = []
        for swz in (1, 0):
            L.relnet_gemm_set_swizzle(swz)
            row.append(timeit(fn))
        L.relnet_gemm_set_swizzle(1)
        print('%-38s XCD-aware order %8.1f us   plain order %8.1f us' % (name, row[0], row[1]), flush=True)


if __name__ == '__main__':
    main()
