"""fp16 against bf16 operands on the same MFMA-tiled kernel (BASELINE configs[4] is worded "fp16 MFMA stress"; this repository runs it with
bf16 operands): the GEMM of one res4 3x3 layer at 54 images (M = 129 276 pixels, N = 256, K = 2304) and of fc_new_1 (M = 16 200 rois,
N = 1024, K = 12 544), fp32 outputs, the SAME tile for both operand types (relnet_gemm_nt forced to tile 2 / 3, relnet_gemm_nt_f16 with the same
tile), operands N(0,1) x small weights.  Prints us per launch and TFLOP/s.      python tools/fp16_rate.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd  # noqa: F401
from relnet_amd import ops, lib

L = lib.load()


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for name, (M, N, K) in (('res4 3x3 as a GEMM, 54 images', (129276, 256, 2304)), ('fc_new_1, 54 x 300 rois', (16200, 1024, 12544))):
    g = torch.Generator().manual_seed(3)
    a32 = torch.randn(M, K, generator=g)
    w32 = torch.randn(N, K, generator=g) * 0.03
    out = torch.empty(M, N, device='cuda', dtype=torch.float32)
    for tile in (2, 3):
        res = {}
        for dt in (torch.bfloat16, torch.float16) * 2:
            a, w = a32.cuda().to(dt), w32.cuda().to(dt)
            if dt == torch.bfloat16:
                L.relnet_gemm_force_tile(tile)
                fn = lambda: ops.gemm_nt(a, w, out=out)
            else:
                fn = lambda: lib.call('relnet_gemm_nt_f16', a.data_ptr(), a.stride(0), w.data_ptr(), w.stride(0), out.data_ptr(), out.stride(0), M, N, K, tile,
                                      torch.cuda.current_stream().cuda_stream)
            us = timeit(fn)
            L.relnet_gemm_force_tile(0)
            res.setdefault(str(dt).split('.')[1], []).append(us)
        line = '  '.join('%s %s us = %s TFLOP/s' % (k, '/'.join('%.1f' % u for u in v), '/'.join('%.0f' % (2.0 * M * N * K / u / 1e6) for u in v)) for k, v in res.items())
        print('%-32s tile %d (%s): %s   fp16 / bf16 time = %.3f' % (name, tile, '256x128' if tile == 2 else '128x128', line,
                                                                     min(res['float16']) / min(res['bfloat16'])))
