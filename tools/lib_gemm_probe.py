"""Probe: what does the vendor GEMM (torch -> hipBLASLt / rocBLAS) reach on the res4 1x1 shapes?  (measurement only)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops
bf = torch.bfloat16
def timeit(fn, name, flops):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 50
    print('%-50s %8.1f us  %6.2f PFLOP/s' % (name, us, flops / us / 1e9))
M = 54 * 38 * 63
for (K, N, nm) in ((1024, 256, 'res4 reduce'), (256, 1024, 'res4 expand'), (2048, 512, 'res5 reduce'), (512, 2048, 'res5 expand'), (2304, 256, 'K=2304 (3x3 as plain GEMM)')):
    x = torch.randn(M, K, device='cuda').to(bf); w = (torch.randn(N, K, device='cuda') * 0.05).to(bf); b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda').to(bf)
    fl = 2.0 * M * N * K
    timeit(lambda: torch.nn.functional.linear(x, w), 'torch linear %s' % nm, fl)
    timeit(lambda: torch.relu(torch.addmm(r, x, w.t())), 'torch addmm + relu %s' % nm, fl)
    if K != 2304:
        timeit(lambda: ops.gemm_nt(x, w, b, relu=True), 'relnet gemm_nt %s' % nm, fl)
        timeit(lambda: ops.gemm_nt(x, w, b, relu=True, resid=r), 'relnet gemm_nt + resid %s' % nm, fl)
