"""Micro-benchmark (not a test): timing ablations of the ring convolution tile (relnet_gemm_debug_ablate).
    python tools/bench_ablate.py compute     fill path alone / LDS + MFMA alone / both (tiles 8 and 16, k order 0 / 1) on the MFMA-bound shapes
    python tools/bench_ablate.py expand      the HBM-bound 1x1 expand convolutions: full / without the shortcut operand / without the output
                                             stores / neither / fill path only (tiles 8, 1, 3)
(one parametrised probe; rounds 2 - 4 kept these as bench_ablate.py / bench_ablate2.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: F401
from bench_tiles import conv_case, timeit, L

mode = sys.argv[1] if len(sys.argv) > 1 else 'compute'
if mode == 'compute':
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
elif mode == 'expand':
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
else:
    raise SystemExit("usage: bench_ablate.py compute|expand")
L.relnet_gemm_force_tile(0)
