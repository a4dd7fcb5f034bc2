"""Micro-benchmark (not a test): relnet_gemm_nt at the GEMM shapes of the detector, per tile.
    python tools/bench_gemm.py plain [B]     TFLOP/s per tile 1 - 5 next to torch.nn.functional.linear (hipBLASLt), B images (default 16)
    python tools/bench_gemm.py resid [B]     residual-add shapes per tile and column tiles per workgroup (row-panel mode), B images (default 27)
(one parametrised probe; rounds 1 - 4 kept these as bench_gemm.py / bench_gemm2.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd  # noqa: F401
from relnet_amd import ops, lib

L = lib.load()


def run(M, N, K, tile, resid=False, iters=10, nloop=0, vendor=False):
    L.relnet_gemm_force_tile(tile); L.relnet_gemm_force_nloop(nloop)
    a = torch.randn(M, K, device='cuda').to(torch.bfloat16)
    w = (torch.randn(N, K, device='cuda') * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device='cuda')
    r = torch.randn(M, N, device='cuda').to(torch.bfloat16) if resid else None
    outs = [torch.empty(M, N, device='cuda', dtype=torch.bfloat16) for _ in range(2)]

    def t(fn):
        for i in range(3):
            fn(i)
        s, e = torch.cuda.Event(True), torch.cuda.Event(True)
        s.record()
        for i in range(iters):
            fn(i)
        e.record(); torch.cuda.synchronize()
        return s.elapsed_time(e) / iters
    ms = t(lambda i: ops.gemm_nt(a, w, b, relu=True, resid=r, out=outs[i & 1]))
    L.relnet_gemm_force_tile(0); L.relnet_gemm_force_nloop(0)
    fl = 2.0 * M * N * K
    line = 'tile%d nloop%d M=%7d N=%5d K=%5d resid=%d  relnet %8.1f us %7.1f TF/s' % (tile, nloop, M, N, K, resid, ms * 1e3, fl / ms / 1e9)
    if resid:
        line += ' %6.2f TB/s' % ((M * K + 2 * M * N) * 2 / 1e9 / ms)
    if vendor:
        mt = t(lambda i: torch.nn.functional.linear(a, w))
        line += ' | hipblaslt %8.1f us %7.1f TF/s' % (mt * 1e3, fl / mt / 1e9)
    print(line)


mode = sys.argv[1] if len(sys.argv) > 1 else 'plain'
if mode == 'plain':
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    P4 = B * 38 * 63
    for (M, N, K) in [(P4, 256, 1024), (P4, 1024, 256), (P4, 256, 2304), (P4, 512, 1024), (P4, 2048, 512), (P4, 512, 2048), (P4, 512, 4608),
                      (B * 75 * 125, 128, 512), (B * 75 * 125, 512, 128), (B * 75 * 125, 128, 1152),
                      (B * 150 * 250, 64, 256), (B * 150 * 250, 256, 64), (B * 150 * 250, 64, 576),
                      (B * 300, 1024, 12544), (B * 300, 2048, 1024), (B * 300, 1024, 1024), (4096, 4096, 4096), (8192, 8192, 8192)]:
        for tl in (1, 2, 3, 4, 5):
            run(M, N, K, tl, iters=20, vendor=True)
elif mode == 'resid':
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 27
    for (M, N, K, res) in [(B * 2394, 1024, 256, True), (B * 2394, 2048, 512, True), (B * 9375, 512, 128, True),
                           (B * 2394, 2048, 1024, False), (B * 2394, 512, 2048, False), (B * 2394, 512, 4608, False)]:
        for tl in (1, 3):
            for nl in (1, 2, 4, 8):
                if nl * (256 if tl == 1 else 128) <= N:
                    run(M, N, K, tl, res, nloop=nl)
else:
    raise SystemExit("usage: bench_gemm.py plain|resid [images]")
