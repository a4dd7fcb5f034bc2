"""Micro-benchmark (not a test): fill path alone / LDS+MFMA path alone / both, tile 8, on the MFMA-bound conv shapes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops, lib
from bench_tiles import conv_case, timeit, L
for name, args in (('res4 3x3 256', (38, 63, 256, 256, 3, 1, False)), ('res4 reduce 1024->256', (38, 63, 1024, 256, 1, 1, False)),
                   ('res5 3x3 512 d2', (38, 63, 512, 512, 3, 2, False)), ('rpn 3x3 1024->512', (38, 63, 1024, 512, 3, 1, False))):
    fn = conv_case(*args)
    for tile, ko in ((8, 0), (8, 1), (16, 0), (16, 1)):
        L.relnet_gemm_force_tile(tile); L.relnet_gemm_debug_korder(ko)
        row = []
        for ab in (0, 1, 2):
            L.relnet_gemm_debug_ablate(ab)
            row.append(timeit(fn, 10))
        L.relnet_gemm_debug_ablate(0)
        print('%-26s tile %2d korder %d full %7.1f  fill-only %7.1f  lds+mfma-only %7.1f us' % (name, tile, ko, row[0], row[1], row[2]))
