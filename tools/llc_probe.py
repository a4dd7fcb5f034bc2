#!/usr/bin/env python
"""Infinity-Cache probe: streaming bandwidth of an in-place and an out-of-place elementwise pass over bf16 buffers of growing
size (torch `add`: 16-byte loads / stores, nothing else), plus the expand kernel of the trunk (relnet_bottleneck_chain,
mid = 256) in place and out of place at 27 / 54 images.  What it answers: does a buffer that fits the 256 MB memory-side
cache stay there ACROSS kernels when it is rewritten in place (write-allocate), and what bandwidth does that give?
Prints one JSON object.  Run on the GPU box:  python tools/llc_probe.py
"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(fn, reps=20, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e-3


def main():
    out = {'elementwise': [], 'expand': []}
    for mb in (16, 32, 64, 96, 128, 160, 192, 224, 256, 288, 384, 512, 1024):
        n = mb * (1 << 20) // 2
        x = torch.zeros(n, device='cuda', dtype=torch.bfloat16)
        y = torch.empty_like(x)
        t_in = timed(lambda: x.add_(1.0))
        t_out = timed(lambda: torch.add(x, 1.0, out=y))
        t_pp = timed(lambda: (torch.add(x, 1.0, out=y), torch.add(y, 1.0, out=x)), reps=10) / 2
        out['elementwise'].append({'MB': mb, 'inplace_TBps': 2 * mb * 1.048576e6 / t_in / 1e12,
                                   'out_of_place_same_src_TBps': 2 * mb * 1.048576e6 / t_out / 1e12,
                                   'pingpong_TBps': 2 * mb * 1.048576e6 / t_pp / 1e12})
        del x, y
    import __graft_entry__ as ge
    ge.build()
    import relnet_amd  # noqa: F401
    from relnet_amd import ops
    g = torch.Generator().manual_seed(3)
    w3 = (torch.randn(1024, 256, generator=g) * 0.05).cuda().to(torch.bfloat16)
    w1 = (torch.randn(256, 1024, generator=g) * 0.03).cuda().to(torch.bfloat16)
    w33 = (torch.randn(256, 9 * 256, generator=g) * 0.02).cuda().to(torch.bfloat16)
    b3 = torch.zeros(1024, device='cuda')
    b1 = torch.zeros(256, device='cuda')
    w3f = ops.pack_w_frag(w3)
    for B in (27, 54):
        x = torch.randn(B, 38, 63, 1024, generator=g).cuda().to(torch.bfloat16).relu_()
        m2 = torch.randn(B, 38, 63, 256, generator=g).cuda().to(torch.bfloat16).relu_()
        mb = x.numel() * 2 / 1.048576e6
        t_o = timed(lambda: ops.bottleneck_chain(m2, x, w3f, None, b3, None))
        t_i = timed(lambda: ops.bottleneck_chain(m2, x, w3f, None, b3, None, inplace=True))
        # a whole res4 unit, in place (reduce -> 3x3 -> expand over x) against out of place
        def unit(inplace):
            y = ops.conv2d_nhwc(x, w1, b1, relu=True)
            y = ops.conv2d_nhwc(y, w33, b1, ksize=3, pad=1, relu=True)
            return ops.bottleneck_chain(y, x, w3f, None, b3, None, inplace=inplace)[0]
        t_uo = timed(lambda: unit(False), reps=23)
        t_ui = timed(lambda: unit(True), reps=23)
        t_red = timed(lambda: ops.conv2d_nhwc(x, w1, b1, relu=True))
        y1 = ops.conv2d_nhwc(x, w1, b1, relu=True)
        t_33 = timed(lambda: ops.conv2d_nhwc(y1, w33, b1, ksize=3, pad=1, relu=True))
        out['expand'].append({'images': B, 'x_MB': mb, 'expand_out_of_place_us': t_o * 1e6, 'expand_inplace_us': t_i * 1e6,
                              'unit_out_of_place_us': t_uo * 1e6, 'unit_inplace_us': t_ui * 1e6,
                              'reduce_alone_us': t_red * 1e6, 'conv3x3_alone_us': t_33 * 1e6})
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
