"""Micro-benchmark (not a test): the HBM-bound 1x1 expand convolution, tile 8: full / without the shortcut operand /
without the output stores / neither / fill path only."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops, lib
from bench_tiles import conv_case, timeit, L
for name, args in (('res4 expand 256->1024', (38, 63, 256, 1024, 1, 1)), ('res5 expand 512->2048', (38, 63, 512, 2048, 1, 1)),
                   ('res2 expand 64->256', (150, 250, 64, 256, 1, 1)), ('res4 reduce 1024->256', (38, 63, 1024, 256, 1, 1))):
    for tile in (8, 1, 3):
        L.relnet_gemm_force_tile(tile)
        row = []
        for resid, ab in ((True, 0), (False, 0), (True, 3), (False, 3), (False, 1)):
            if tile != 8 and ab:
                row.append(float('nan')); continue
            fn = conv_case(*args, resid)
            L.relnet_gemm_debug_ablate(ab)
            row.append(timeit(fn, 10))
        L.relnet_gemm_debug_ablate(0)
        print('%-24s tile %d  full %7.1f  no-resid %7.1f  no-store %7.1f  no-resid-no-store %7.1f  fill-only %7.1f us' % ((name, tile) + tuple(row)))
