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
