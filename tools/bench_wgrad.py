"""Micro-benchmark (not a test): the weight-gradient kernel (csrc/wgrad.hip) on the layer shapes of the training step at B images,
under different workgroup targets / tile heights / ablation modes.   python tools/bench_wgrad.py [B]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd  # noqa: F401
from relnet_amd import ops, lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
L = lib.load()


def timeit(fn, iters=6):
    for _ in range(2):
        fn()
    s, e = torch.cuda.Event(True), torch.cuda.Event(True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


def case(H, W, Cin, Cout, k, dil=1, stride=1):
    x = torch.randn(B, H, W, Cin, device='cuda').to(torch.bfloat16)
    pad = dil if k == 3 else 0
    Ho, Wo = (H + 2 * pad - dil * (k - 1) - 1) // stride + 1, (W + 2 * pad - dil * (k - 1) - 1) // stride + 1
    dy = torch.randn(B * Ho * Wo, Cout, device='cuda').to(torch.bfloat16)
    out = torch.zeros(Cout, k * k * Cin, device='cuda')
    conv = None if (k == 1 and stride == 1) else (k, stride, dil, pad)
    xx = x.view(-1, Cin) if conv is None else x
    return (dy, xx, out, None, None, conv), 2.0 * dy.shape[0] * Cout * k * k * Cin


def fc_case(P, K, Cout):
    x = torch.randn(P, K, device='cuda').to(torch.bfloat16)
    dy = torch.randn(P, Cout, device='cuda').to(torch.bfloat16)
    out = torch.zeros(Cout, K, device='cuda')
    return (dy, x, out, None, None, None), 2.0 * P * K * Cout


def unit(h, w, cin, mid, dil=1, proj=None):
    """the three (four) convolutions of one bottleneck unit: reduce, 3x3, expand (+ projection shortcut)"""
    r = [case(h, w, cin, mid, 1), case(h, w, mid, mid, 3, dil), case(h, w, mid, 4 * mid, 1)]
    if proj:
        r.append(case(h, w, proj, 4 * mid, 1))
    return r


GROUPS = [
    ('res4 (23 units, 69 layers)', lambda: sum([unit(38, 63, 1024, 256) for _ in range(23)], [])),
    ('res5 (3 units + proj)', lambda: unit(38, 63, 1024, 512, 2, proj=1024) + unit(38, 63, 2048, 512, 2) + unit(38, 63, 2048, 512, 2)),
    ('res3 (4 units)', lambda: sum([unit(75, 125, 512, 128) for _ in range(4)], [])),
    ('heads (rpn 3x3, conv_new_1, fc_new_1, 3 fc, 2 qk)', lambda: [case(38, 63, 1024, 512, 3), case(38, 63, 2048, 256, 1), fc_case(B * 308, 12544, 1024)] +
     [fc_case(B * 308, 1024, 1024) for _ in range(3)] + [fc_case(B * 308, 1024, 2048) for _ in range(2)]),
    ('one layer: res4 3x3', lambda: [case(38, 63, 256, 256, 3)]),
    ('one layer: fc_new_1', lambda: [fc_case(B * 308, 12544, 1024)]),
]
SETTINGS = [('256wg', 256, 0, 0), ('512wg', 512, 0, 0), ('no flush', 256, 1, 0), ('no flush no loads', 256, 5, 0), ('wm2', 256, 0, 2)]


def main():
    tot = {s[0]: 0.0 for s in SETTINGS}
    tf = 0.0
    for name, mk in GROUPS:
        items = mk()
        flops = sum(f for _, f in items)

        def fn():
            q = ops.WgradQueue()
            for a, _ in items:
                q.add(*a)
            q.flush()
        row = []
        for label, blocks, mode, wm in SETTINGS:
            L.relnet_wgrad_tune(blocks, mode, wm)
            us = timeit(fn)
            row.append(us)
            if not name.startswith('one layer'):
                tot[label] += us
        L.relnet_wgrad_tune(0, 0, 0)
        if not name.startswith('one layer'):
            tf += flops
        print('%-52s %7.1f GF  ' % (name, flops / 1e9) + ' '.join('%8.1f' % r for r in row) + '   us   %.2f PF/s' % (flops / row[0] / 1e9))
    print('settings: ' + ' | '.join(s[0] for s in SETTINGS))
    print('per-step totals (ms): ' + ' | '.join('%s %.2f' % (k, v / 1e3) for k, v in tot.items()) + '   total %.2f TFLOP' % (tf / 1e12))


if __name__ == '__main__':
    main()
