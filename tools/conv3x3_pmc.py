"""Launch only the res4 3x3 convolution of the default bench (54 images) -- under rocprofv3 --pmc, with TILE / ABLATE from the
environment (ABLATE 1 = fill path only, 2 = LDS + MFMA only: relnet_gemm_debug_ablate):  python tools/conv3x3_pmc.py [iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import relnet_amd
from relnet_amd import ops, lib
L = lib.load()
L.relnet_gemm_force_tile(int(os.environ.get('TILE', '8')))
L.relnet_gemm_debug_ablate(int(os.environ.get('ABLATE', '0')))
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 4
x = torch.randn(54, 38, 63, 256, device='cuda').to(torch.bfloat16)
w = (torch.randn(256, 9 * 256, device='cuda') * 0.03).to(torch.bfloat16)
b = torch.randn(256, device='cuda')
for _ in range(iters):
    z = ops.conv2d_nhwc(x, w, b, ksize=3, pad=1, relu=True)
torch.cuda.synchronize()
print('done')
