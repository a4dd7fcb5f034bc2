"""Time relnet_conv3x3_c64 / relnet_bottleneck_chain alone at the bench shape (graph replay)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 54
bf = torch.bfloat16
x = torch.relu(torch.randn(B, 150, 250, 64, device='cuda')).to(bf)
w = (torch.randn(64, 64, 3, 3) * 0.05).to(bf)
b = torch.randn(64, device='cuda') * 0.1
wp = ops.pack_conv_weight(w, bf, 'cuda')
wf = ops.pack_w_frag(wp, panel_only=False)
def timeit(fn, name):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        with torch.cuda.graph(g, stream=st):
            for _ in range(10):
                fn()
    g.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    print('%s: %.1f us' % (name, e0.elapsed_time(e1) * 1e3 / 50))
timeit(lambda: ops.conv3x3_c64(x, wf, b), 'halo 3x3')
if True:
    timeit(lambda: ops.conv2d_nhwc(x, wp, b, ksize=3, pad=1, relu=True), 'implicit gemm')
for mid, (H, W) in ((64, (150, 250)), (128, (75, 125))):
    m2 = torch.relu(torch.randn(B, H, W, mid, device='cuda')).to(bf)
    xx = torch.relu(torch.randn(B, H, W, 4 * mid, device='cuda')).to(bf)
    w3 = (torch.randn(4 * mid, mid, device='cuda') * 0.1).to(bf); w1 = (torch.randn(mid, 4 * mid, device='cuda') * 0.05).to(bf)
    b3 = torch.randn(4 * mid, device='cuda') * 0.1; b1 = torch.randn(mid, device='cuda') * 0.1
    w3f, w1f = ops.pack_w_frag(w3), ops.pack_chain_w1(w1)
    gb = m2.numel() * 2 * 2 * 5 / 1e9
    timeit(lambda: ops.bottleneck_chain(m2, xx, w3f, w1f, b3, b1), 'chain mid=%d (%.2f GB)' % (mid, gb))
    def two():
        xn = ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=xx)
        return ops.conv2d_nhwc(xn, w1, b1, relu=True)
    timeit(two, 'two launches mid=%d' % mid)
# res4 expand + shortcut: expand-only chain vs the row-panel kernel
mid, (H, W) = 256, (38, 63)
m2 = torch.relu(torch.randn(B, H, W, mid, device='cuda')).to(bf)
xx = torch.relu(torch.randn(B, H, W, 4 * mid, device='cuda')).to(bf)
w3 = (torch.randn(4 * mid, mid, device='cuda') * 0.05).to(bf); b3 = torch.randn(4 * mid, device='cuda') * 0.1
w3f = ops.pack_w_frag(w3)
timeit(lambda: ops.bottleneck_chain(m2, xx, w3f, None, b3, None), 'chain expand-only mid=256')
timeit(lambda: ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=xx, w_frag=w3f), 'row-panel kernel (tile 14)')
timeit(lambda: ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=xx), 'tile 1 (256x256)')
a_, _ = ops.bottleneck_chain(m2, xx, w3f, None, b3, None)
b_ = ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=xx)
print('max diff vs conv', (a_.float() - b_.float()).abs().max().item(), 'differs in %.4f%%' % (100 * (a_ != b_).float().mean().item()))
mid, (H, W) = 512, (38, 63)
m2 = torch.relu(torch.randn(B, H, W, mid, device='cuda')).to(bf)
xx = torch.relu(torch.randn(B, H, W, 4 * mid, device='cuda')).to(bf)
w3 = (torch.randn(4 * mid, mid, device='cuda') * 0.05).to(bf); b3 = torch.randn(4 * mid, device='cuda') * 0.1
w3f = ops.pack_w_frag(w3)
timeit(lambda: ops.bottleneck_chain(m2, xx, w3f, None, b3, None), 'chain expand-only mid=512')
timeit(lambda: ops.conv2d_nhwc(m2, w3, b3, relu=True, resid=xx), 'tile 1 (256x256) mid=512')
