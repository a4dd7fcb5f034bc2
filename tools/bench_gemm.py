"""Micro-benchmark (not a test): TFLOP/s of relnet_gemm_nt at the shapes of the detector."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops

def bench(M, N, K, dtype=torch.bfloat16, iters=20, relu=True, tile=0):
    from relnet_amd import lib as L
    L.load().relnet_gemm_force_tile(tile)
    a = torch.randn(M, K, device='cuda').to(dtype)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(dtype)
    b = torch.randn(N, device='cuda')
    out = torch.empty(M, N, device='cuda', dtype=dtype)
    for _ in range(3):
        ops.gemm_nt(a, w, b, relu=relu, out=out)
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        ops.gemm_nt(a, w, b, relu=relu, out=out)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / iters
    for _ in range(3):
        torch.nn.functional.linear(a, w)
    s.record()
    for _ in range(iters):
        torch.nn.functional.linear(a, w)
    e.record(); torch.cuda.synchronize()
    ms_t = s.elapsed_time(e) / iters
    fl = 2.0 * M * N * K
    print('tile%d M=%7d N=%5d K=%5d  relnet %8.1f us %7.1f TF/s | hipblaslt %8.1f us %7.1f TF/s' % (tile, M, N, K, ms * 1e3, fl / ms / 1e9, ms_t * 1e3, fl / ms_t / 1e9))

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
P4 = B * 38 * 63
for (M, N, K) in [(P4, 256, 1024), (P4, 1024, 256), (P4, 256, 2304), (P4, 512, 1024), (P4, 2048, 512), (P4, 512, 2048), (P4, 512, 4608),
                  (B * 75 * 125, 128, 512), (B * 75 * 125, 512, 128), (B * 75 * 125, 128, 1152),
                  (B * 150 * 250, 64, 256), (B * 150 * 250, 256, 64), (B * 150 * 250, 64, 576),
                  (B * 300, 1024, 12544), (B * 300, 2048, 1024), (B * 300, 1024, 1024), (4096, 4096, 4096), (8192, 8192, 8192)]:
    for t in (1, 2, 3, 4, 5):
        bench(M, N, K, tile=t)
