"""Micro-benchmark (not a test): residual-add GEMM shapes per tile config."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops, lib

def bench(M, N, K, tile, resid=True, iters=10, nloop=1):
    lib.load().relnet_gemm_force_tile(tile)
    lib.load().relnet_gemm_force_nloop(nloop)
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda').to(torch.bfloat16) if resid else None
    outs = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(2)]
    for i in range(2):
        ops.gemm_nt(a, w, b, relu=True, resid=r, out=outs[i & 1])
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for i in range(iters):
        ops.gemm_nt(a, w, b, relu=True, resid=r, out=outs[i & 1])
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    gb = (M * K + M * N * (2 if resid else 1)) * 2 / 1e9
    print('nloop%d tile%d M=%7d N=%5d K=%5d resid=%d %8.1f us %7.1f TF/s %6.2f TB/s' % (nloop, tile, M, N, K, resid, ms * 1e3, 2.0 * M * N * K / ms / 1e9, gb / ms))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 27
for (M, N, K, res) in [(B * 2394, 1024, 256, True), (B * 2394, 2048, 512, True), (B * 9375, 512, 128, True),
                       (B * 2394, 2048, 1024, False), (B * 2394, 512, 2048, False), (B * 2394, 512, 4608, False)]:
    for t in (1, 3):
        for nl in (1, 2, 4, 8):
            if nl * (256 if t == 1 else 128) <= N:
                bench(M, N, K, t, res, nloop=nl)
