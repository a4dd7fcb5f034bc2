#!/usr/bin/env python
"""Where a workgroup of the 256 x 256 ring tiles spends its life (relnet_gemm_debug_phase_ts): per layer shape at 54 images the
set-up (address arithmetic + first loads issued), the k-loop, the epilogue, and the SHADER CLOCK the k-loop actually ran at
(s_memtime cycles / s_memrealtime wall time) -- the dense-MFMA peak quoted at 2.4 GHz is not the clock these kernels get.
python tools/tile_phase_probe.py [images]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: F401,E402
from relnet_amd import ops, lib  # noqa: E402

L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 54


def case(H, W, Cin, Cout, k, dil):
    x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.03).to(torch.bfloat16)
    b = torch.randn(Cout, device='cuda')
    out = torch.empty(B, H, W, Cout, device='cuda', dtype=torch.bfloat16)
    return lambda: ops.conv2d_nhwc(x, w, b, ksize=k, pad=dil if k == 3 else 0, dil=dil, relu=True, out=out)


def main():
    for name, args in (('res4 3x3 256', (38, 63, 256, 256, 3, 1)), ('res4 reduce 1024->256', (38, 63, 1024, 256, 1, 1)),
                       ('res5 3x3 512 d2', (38, 63, 512, 512, 3, 2)), ('rpn 3x3 1024->512', (38, 63, 1024, 512, 3, 1))):
        fn = case(*args)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        nwg = ((B * 38 * 63 + 255) // 256) * (args[3] // 256)
        ts = torch.zeros(nwg * 8, device='cuda', dtype=torch.int64)
        L.relnet_gemm_debug_phase_ts(ts.data_ptr())
        fn()
        torch.cuda.synchronize()
        L.relnet_gemm_debug_phase_ts(None)
        raw = ts.view(nwg, 8).cpu().double()
        t = (raw[:, :4] - raw[:, 0].min()) * 0.01                  # 100 MHz ticks -> us
        nslab = args[2] * args[4] * args[4] // 64
        mhz = (raw[:, 4] / ((raw[:, 2] - raw[:, 0]) * 0.01)).mean().item()
        setup, loop, epi = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
        print('%-22s %4d workgroups, launch span %6.1f us | set-up %.2f us, k-loop %.2f us (%d slabs: %.3f us = %.0f cycles each), '
              'epilogue %.2f us | shader clock %.0f MHz -> MFMA busy %.0f %% of the k-loop'
              % (name, nwg, t[:, 3].max().item(), setup.mean().item(), loop.mean().item(), nslab, loop.mean().item() / nslab,
                 loop.mean().item() / nslab * mhz, epi.mean().item(), mhz, 100.0 * 2048.0 / (loop.mean().item() / nslab * mhz)), flush=True)


if __name__ == '__main__':
    main()
