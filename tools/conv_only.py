"""Launch the three res4 convolution shapes of the default bench (54 images) a few times -- used under rocprofv3 --pmc to
read HBM traffic and MFMA utilisation of the implicit-GEMM kernel:  python tools/conv_only.py [B] [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops, lib
lib.load().relnet_gemm_force_tile(int(os.environ.get('TILE', '0')))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 54
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 4
H, W = 38, 63
x256 = torch.randn(B, H, W, 256, device='cuda').to(torch.bfloat16)
x1024 = torch.randn(B, H, W, 1024, device='cuda').to(torch.bfloat16)
w_exp = (torch.randn(1024, 256, device='cuda') * 0.03).to(torch.bfloat16)
w_3x3 = (torch.randn(256, 9 * 256, device='cuda') * 0.03).to(torch.bfloat16)
w_red = (torch.randn(256, 1024, device='cuda') * 0.03).to(torch.bfloat16)
b1024, b256 = torch.randn(1024, device='cuda'), torch.randn(256, device='cuda')
for _ in range(iters):
    y = ops.conv2d_nhwc(x256, w_exp, b1024, ksize=1, relu=True, resid=x1024)          # expand + residual + ReLU
    z = ops.conv2d_nhwc(x256, w_3x3, b256, ksize=3, pad=1, relu=True)                  # 3x3
    r = ops.conv2d_nhwc(x1024, w_red, b256, ksize=1, relu=True)                        # reduce
torch.cuda.synchronize()
print('done')
