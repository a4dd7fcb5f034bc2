import torch, time, sys
sys.path.insert(0, '/root/repo')
import relnet_amd
from relnet_amd import ops, lib
L = lib.load()
def timeit(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
g = torch.Generator().manual_seed(0)
shapes = [('res4_3x3', (38,63), 256, 256, 3, 1), ('res4_reduce', (38,63), 1024, 256, 1, 1), ('res4_expand', (38,63), 256, 1024, 1, 1),
          ('res5_3x3', (38,63), 512, 512, 3, 2), ('res5_reduce', (38,63), 2048, 512, 1, 1), ('res5_expand', (38,63), 512, 2048, 1, 1), ('rpn_3x3', (38,63), 1024, 512, 3, 1),
          ('res3_3x3', (75,125), 128, 128, 3, 1), ('res3_reduce', (75,125), 512, 128, 1, 1), ('conv_new_1', (38,63), 2048, 256, 1, 1)]
for name, hw, cin, cout, k, dil in shapes:
    x = torch.randn(1, hw[0], hw[1], cin, generator=g).cuda().to(torch.bfloat16)
    wp = (torch.randn(cout, cin*k*k, generator=g) / (cin*k*k) ** 0.5).cuda().to(torch.bfloat16)
    bias = torch.randn(cout, generator=g).cuda()
    f = lambda: ops.conv2d_nhwc(x, wp, bias, ksize=k, pad=dil*(k//2), dil=dil, relu=True)
    row = [name, 'auto-tile', L.relnet_gemm_pick_tile(hw[0]*hw[1], cout, cin*k*k, 1, 1)]
    for ways in (1, 0, 2, 3, 4, 6, 8):
        L.relnet_gemm_debug_splitk(ways)
        if ways >= 2: L.relnet_gemm_force_tile(23)
        row.append('%s:%.1f' % ({1: 'off', 0: 'auto'}.get(ways, ways), timeit(f)))
        L.relnet_gemm_force_tile(0)
    L.relnet_gemm_debug_splitk(0)
    print(*row)
for name, M, N, K in [('fc_new_1', 300, 1024, 12544), ('fc_new_2', 300, 1024, 1024), ('qk', 300, 2048, 1024), ('fc1_dgrad', 308, 12544, 1024)]:
    a = torch.randn(M, K, generator=g).cuda().to(torch.bfloat16)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().to(torch.bfloat16)
    b = torch.randn(N, generator=g).cuda()
    f = lambda: ops.gemm_nt(a, w, b, relu=True)
    row = [name, 'auto-tile', L.relnet_gemm_pick_tile(M, N, K, 1, 1)]
    for ways in (1, 0, 2, 4, 6, 8):
        L.relnet_gemm_debug_splitk(ways)
        if ways >= 2: L.relnet_gemm_force_tile(23)
        row.append('%s:%.1f' % ({1: 'off', 0: 'auto'}.get(ways, ways), timeit(f)))
        L.relnet_gemm_force_tile(0)
    L.relnet_gemm_debug_splitk(0)
    print(*row)
