"""Micro-benchmark (not a test): XCD-aware tile order on / off (and n_loop 1 vs auto), per convolution shape.
python tools/bench_swizzle.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench_tiles as BT
L = BT.L
tot = {}
for name, cnt, args in BT.CASES:
    fn = BT.conv_case(*args)
    row = {}
    for sw in (0, 1):
        for nl in (0, 1):
            L.relnet_gemm_set_swizzle(sw); L.relnet_gemm_force_nloop(nl)
            row[(sw, nl)] = BT.timeit(fn)
            tot[(sw, nl)] = tot.get((sw, nl), 0.0) + cnt * row[(sw, nl)]
    L.relnet_gemm_set_swizzle(1); L.relnet_gemm_force_nloop(0)
    print('%-30s x%2d  plain/auto %7.1f  plain/nloop1 %7.1f | swizzle/auto %7.1f  swizzle/nloop1 %7.1f us' % (
        name, cnt, row[(0, 0)], row[(0, 1)], row[(1, 0)], row[(1, 1)]))
print('per-step totals (ms): plain/auto %.2f  plain/nloop1 %.2f  swizzle/auto %.2f  swizzle/nloop1 %.2f' % (
    tot[(0, 0)] / 1e3, tot[(0, 1)] / 1e3, tot[(1, 0)] / 1e3, tot[(1, 1)] / 1e3))
