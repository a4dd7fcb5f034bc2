"""relnet_deformable_col2im on the res5 layer of the 8-image DCN training step: the atomic scatter kernel (mode 1) against the gather + offset
kernels (mode 0) at small / medium / large offsets, fp32 and bf16 column gradients.   python tools/col2im_probe.py [B]"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: E402,F401
from relnet_amd import ops, lib  # noqa: E402
from relnet_amd.ops import _strides4, _dt, _stream, F32, BF16  # noqa: E402
L = lib.load()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
C, H, W, dg = 512, 38, 63, 4
g = torch.Generator().manual_seed(0)
x = torch.randn(B, C, H, W, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last)
for sigma in (0.05, 0.5, 1.5, 4.0):
    o = (torch.randn(B, 18 * dg, H, W, generator=g) * sigma).cuda()
    for cdt in (torch.float32, torch.bfloat16):
        dcol = torch.randn(B * H * W, 9 * C, generator=g).cuda().to(cdt)
        gd = torch.zeros(B, H, W, C, device='cuda').permute(0, 3, 1, 2)
        go = torch.zeros(B, H, W, 18 * dg, device='cuda').permute(0, 3, 1, 2)
        row = ['sigma %.2f dcol %s' % (sigma, 'f32' if cdt == torch.float32 else 'bf16')]
        for mode in (1, 11, 12, 13):
            L.relnet_deformable_col2im_debug(mode)
            def f():
                lib.call('relnet_deformable_col2im', dcol.data_ptr(), dcol.stride(0), F32 if cdt == torch.float32 else BF16, x.data_ptr(), _strides4(x), _dt(x),
                         o.data_ptr(), _strides4(o), gd.data_ptr(), _strides4(gd), go.data_ptr(), _strides4(go), B, C, H, W, 3, 3, 2, 2, 1, 1, 2, 2, dg, _stream())
            for _ in range(3): f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): f()
            e1.record(); torch.cuda.synchronize()
            row.append('%s %.1f us' % ({1: 'scatter'}.get(mode, 'gather D=%d' % (mode - 10)), e0.elapsed_time(e1) / 20 * 1e3))
        L.relnet_deformable_col2im_debug(0)
        print(*row)
