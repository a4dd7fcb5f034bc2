"""Launch ONE of the trunk kernels of the default bench in isolation, a few times, for `rocprofv3 --pmc` (round 5: counter evidence for the
hand-scheduled ring tile and the role-specialised res4 chain kernel).
    python tools/kernel_pmc.py <what> [images] [iters]
what = res4_3x3   3x3 / 256 -> 256 convolution of a res4 unit on the asm ring tile (tile 19)              152.5 GFLOP per launch at 54 images
       res5_3x3   3x3 dilated / 512 -> 512 convolution of a res5 unit (tile 19)                            610.0 GFLOP
       res4_reduce  1x1 / 1024 -> 256 (tile 19; inside the chain kernel in the step, here for the HBM-fed case)
       chain256   relnet_bottleneck_chain, mid 256: expand 256 -> 1024 + shortcut + ReLU and the next unit's reduce 1024 -> 256 + ReLU
                  (chain256_roles_kernel), out of place
TILE=<t> forces another tile for the convolution cases."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd  # noqa: F401
from relnet_amd import ops, lib

what = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 54
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 4
L = lib.load()
H, W = 38, 63
g = torch.Generator().manual_seed(1)
rn = lambda *s: torch.randn(*s, generator=g).cuda()
if what in ('res4_3x3', 'res5_3x3', 'res4_reduce'):
    cin, cout, k, dil = {'res4_3x3': (256, 256, 3, 1), 'res5_3x3': (512, 512, 3, 2), 'res4_reduce': (1024, 256, 1, 1)}[what]
    x = rn(B, H, W, cin).to(torch.bfloat16)
    w = (rn(cout, k * k * cin) * 0.03).to(torch.bfloat16)
    b = rn(cout)
    L.relnet_gemm_force_tile(int(os.environ.get('TILE', '19')))
    run = lambda: ops.conv2d_nhwc(x, w, b, ksize=k, pad=dil if k == 3 else 0, dil=dil, relu=True)
elif what == 'chain256':
    P = B * H * W
    m2 = rn(P, 256).to(torch.bfloat16)
    xs = rn(P, 1024).to(torch.bfloat16)
    w3 = ops.pack_w_frag((rn(1024, 256) * 0.05).to(torch.bfloat16))
    w1 = ops.pack_chain_w1((rn(256, 1024) * 0.03).to(torch.bfloat16))
    b3, b1 = rn(1024), rn(256)
    L.relnet_chain_debug(int(os.environ.get('CHAIN_DBG', '0')))       # timing ablations (see include/relnet_hip.h)
    run = lambda: ops.bottleneck_chain(m2, xs, w3, w1, b3, b1, inplace=False)
else:
    raise SystemExit('unknown kernel %r' % what)
for _ in range(iters):
    run()
torch.cuda.synchronize()
if os.environ.get('TIME'):
    s_, e_ = torch.cuda.Event(True), torch.cuda.Event(True)
    s_.record()
    for _ in range(20):
        run()
    e_.record(); torch.cuda.synchronize()
    print('TIME %s B=%d dbg=%s: %.1f us per launch' % (what, B, os.environ.get('CHAIN_DBG', '0'), s_.elapsed_time(e_) / 20 * 1e3))
print('done', what, B, iters)
