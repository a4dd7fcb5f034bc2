"""Micro-benchmark (not a test): every convolution shape of the detector at B images, per tile configuration.
python tools/bench_tiles.py [B]   -> prints the time of tile configs 0 (auto) / 1 / 2 / 3 / 4 per shape."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 54
L = lib.load()


def timeit(fn, iters=8):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def conv_case(H, W, Cin, Cout, k, dil, resid, stride=1):
    x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
    w = (torch.randn(Cout, k * k * Cin, device='cuda') * 0.03).to(torch.bfloat16)
    if os.environ.get('ZERO_OPERANDS'):          # data-dependent power: the same launches on all-zero operands
        x.zero_(); w.zero_()
    b = torch.randn(Cout, device='cuda')
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    r = torch.randn(B, Ho, Wo, Cout, device='cuda').to(torch.bfloat16) if resid else None
    out = torch.empty(B, Ho, Wo, Cout, device='cuda', dtype=torch.bfloat16)
    wf = ops.pack_w_frag(w) if (k == 1 and stride == 1) else None
    return lambda: ops.conv2d_nhwc(x, w, b, ksize=k, stride=stride, pad=dil if k == 3 else 0, dil=dil, relu=True, resid=r, out=out, w_frag=wf)


CASES = [  # name, count per step, args
    ('res4 expand 256->1024 +res', 23, (38, 63, 256, 1024, 1, 1, True)),
    ('res4 3x3 256', 23, (38, 63, 256, 256, 3, 1, False)),
    ('res4 reduce 1024->256', 22, (38, 63, 1024, 256, 1, 1, False)),
    ('res2 expand 64->256 +res', 3, (150, 250, 64, 256, 1, 1, True)),
    ('res5 3x3 512 d2', 3, (38, 63, 512, 512, 3, 2, False)),
    ('res5 expand 512->2048 +res', 3, (38, 63, 512, 2048, 1, 1, True)),
    ('res3 expand 128->512 +res', 4, (75, 125, 128, 512, 1, 1, True)),
    ('rpn 3x3 1024->512', 1, (38, 63, 1024, 512, 3, 1, False)),
    ('res3 3x3 128', 4, (75, 125, 128, 128, 3, 1, False)),
    ('res2 3x3 64', 3, (150, 250, 64, 64, 3, 1, False)),
    ('res5 reduce 2048->512', 2, (38, 63, 2048, 512, 1, 1, False)),
    ('res3 reduce 512->128', 3, (75, 125, 512, 128, 1, 1, False)),
    ('res2 reduce 256->64', 2, (150, 250, 256, 64, 1, 1, False)),
    ('conv_new_1 2048->256', 1, (38, 63, 2048, 256, 1, 1, False)),
]
if os.environ.get('TRAIN_CASES'):      # the data-gradient shapes of the training step (same kernels: a 3x3 / 1x1 convolution of dY with a shortcut or mask operand)
    CASES += [
        ('res4 3x3 dgrad +mask', 23, (38, 63, 256, 256, 3, 1, True)),
        ('res4 expand-dgrad 1024->256 +mask', 23, (38, 63, 1024, 256, 1, 1, True)),
        ('res4 reduce-dgrad 256->1024 +res', 23, (38, 63, 256, 1024, 1, 1, True)),
        ('res3 3x3 dgrad +mask', 4, (75, 125, 128, 128, 3, 1, True)),
        ('res3 expand-dgrad 512->128 +mask', 4, (75, 125, 512, 128, 1, 1, True)),
        ('res5 3x3 dgrad +mask d2', 3, (38, 63, 512, 512, 3, 2, True)),
        ('res5 expand-dgrad 2048->512 +mask', 3, (38, 63, 2048, 512, 1, 1, True)),
        ('res5 reduce-dgrad 512->2048 +res', 3, (38, 63, 512, 2048, 1, 1, True)),
        ('rpn 3x3 dgrad 512->1024', 1, (38, 63, 512, 1024, 3, 1, False)),
    ]
TILES = [int(x) for x in os.environ.get('TILES', '0,1,8,14').split(',')]


def main():
    tot = {t: 0.0 for t in TILES}
    best_tot = 0.0
    for name, cnt, args in CASES:
        fn = conv_case(*args)
        row = []
        for t in TILES:
            L.relnet_gemm_force_tile(t)
            try:
                us = timeit(fn)
            except Exception as ex:
                us = float('nan')
            row.append(us); tot[t] += cnt * us
        L.relnet_gemm_force_tile(0)
        best_tot += cnt * min(r for r in row if r == r)
        bi = min(range(len(row)), key=lambda i: row[i] if row[i] == row[i] else 1e30)
        print('%-30s x%2d ' % (name, cnt) + ' '.join('t%d %7.1f' % (t, r) for t, r in zip(TILES, row)) + '  us   best t%d' % TILES[bi])
    print('per-step totals (ms): ' + ' '.join('t%d %.2f' % (t, tot[t] / 1e3) for t in TILES) + '   per-shape best %.2f' % (best_tot / 1e3))


if __name__ == '__main__':
    main()
