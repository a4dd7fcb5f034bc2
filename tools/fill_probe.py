#!/usr/bin/env python
"""L2 -> LDS fill-path probe (round 4): does the fill rate of the 256 x 256 x 64 ring tile depend on the ROW STRIDES of its
operands (cache-channel hot spots)?  1x1 convolution M = 54 x 38 x 63 pixels, N = 256, K = Cin, for several Cin and padded
activation pixel strides; per case the full kernel (tiles 8 / 18) and the fill-only ablation of tile 8.
python tools/fill_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: F401,E402
from relnet_amd import ops, lib  # noqa: E402

L = lib.load()


def timeit(fn, iters=10):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def main():
    B, H, W, N = 54, 38, 63, 256
    print('%-28s %10s %10s %12s %14s' % ('case', 'tile8 us', 'tile18 us', 'fill-only us', 'fill TB/s'))
    for cin in (1024, 1088, 2304, 2368, 2432):
        for pad in (0, 64):
            xb = torch.randn(B, H, W, cin + pad, device='cuda').to(torch.bfloat16)
            x = xb[..., :cin]
            w = (torch.randn(N, cin, device='cuda') * 0.03).to(torch.bfloat16)
            b = torch.zeros(N, device='cuda')
            out = torch.empty(B, H, W, N, device='cuda', dtype=torch.bfloat16)
            fn = lambda: ops.conv2d_nhwc(x, w, b, ksize=1, relu=True, out=out)
            res = {}
            for t in (8, 18):
                L.relnet_gemm_force_tile(t)
                res[t] = timeit(fn)
            L.relnet_gemm_force_tile(8); L.relnet_gemm_debug_ablate(1)
            fo = timeit(fn)
            L.relnet_gemm_debug_ablate(0); L.relnet_gemm_force_tile(0)
            M = B * H * W
            tiles = (M + 255) // 256
            fill_bytes = tiles * (cin // 64) * 65536          # every tile stages one 32 KB activation slab + one 32 KB filter slab per 64-deep k step
            print('Cin %4d pixel stride %4d   %10.1f %10.1f %12.1f %14.2f' % (cin, cin + pad, res[8], res[18], fo, fill_bytes / fo / 1e6))


if __name__ == '__main__':
    main()
