"""ROI pooling backward of the 8-image training step in isolation: owner form (mode 0) against the scatter kernel (1) and the owner kernel's timing
ablations (relnet_roi_pool_bwd_debug(4 + bits): 1 = no LDS adds, 2 = no scattered loads; results wrong) -> profiles/r06_notes/ab_one_image_step.txt (21).
    python tools/roi_bwd_probe.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd
from relnet_amd import ops, lib
L = lib.load()
B, R, C, H, W = 8, 308, 256, 38, 63
g = torch.Generator().manual_seed(0)
feat = torch.randn(B, H, W, C, generator=g).cuda().to(torch.bfloat16)
import numpy as np
rng = np.random.default_rng(0)
rois = []
for b in range(B):
    bw, bh = rng.uniform(32, 500, R), rng.uniform(32, 400, R)
    x1, y1 = rng.uniform(0, 999 - bw), rng.uniform(0, 599 - bh)
    rois.append(np.stack([np.full(R, b), x1, y1, x1 + bw, y1 + bh], 1))
rois = torch.as_tensor(np.concatenate(rois).astype(np.float32)).cuda()
pooled, argmax = ops.roi_pool(feat.permute(0, 3, 1, 2), rois, (7, 7), 1 / 16.0, channels_last_out=True, want_argmax=True)
dy = torch.randn(pooled.shape, generator=g).cuda().to(torch.bfloat16).contiguous(memory_format=torch.channels_last) if False else torch.randn_like(pooled)
def run():
    return ops.roi_pool_bwd(dy, argmax, rois, (B, C, H, W), channels_last=True)
for mode in (0, 1, 4, 5, 6, 7):
    L.relnet_roi_pool_bwd_debug(mode)
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): run()
    e1.record(); torch.cuda.synchronize()
    print('mode', mode, '%.1f us per call (incl. the zero fill)' % (e0.elapsed_time(e1) / 50 * 1e3), flush=True)
L.relnet_roi_pool_bwd_debug(0)
