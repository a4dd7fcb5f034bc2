"""Several batches in flight: does replaying n captured steps (n Detector instances, n streams) round-robin beat one step after the other?
    python tools/pipeline_probe.py [images per step] [n]
Prints images/s of (a) one graph replayed back to back, (b) n graphs replayed round-robin on n streams.  Each graph is captured on ITS
stream: graphs captured on one stream share the runtime's queues and do not overlap (measured: +0 - 1 % instead of +2 - 5 %).
PROBE_H2D=1: every step first uploads its batch from pinned host memory into the slot's resident input tensor on the slot's stream (the
PCIe-inclusive rate of DESIGN.md section 6)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: E402,F401
from relnet_amd import backbone, detector  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 54
NG = int(sys.argv[2]) if len(sys.argv) > 2 else 2
params = backbone.init_params(seed=1)
g = torch.Generator().manual_seed(1000)
im_info = torch.tensor([[600.0, 1000.0, 1.0]] * B).cuda()
steps, keep, hosts = [], [], []
H2D = bool(os.environ.get('PROBE_H2D'))
with torch.no_grad():
    for i in range(NG):
        det = detector.Detector(params, dtype=torch.bfloat16, device='cuda', relation=True, cfg=detector.Config())
        if os.environ.get('PROBE_NO_RPN_STREAM'):      # one queue per step: the RPN branch in line
            det.overlap_rpn = False
        data = torch.randn(B, 3, 600, 1000, generator=g).cuda()
        keep.append((det, data))
        hosts.append(torch.randn(B, 3, 600, 1000, generator=g).pin_memory() if H2D else None)
        steps.append(lambda det=det, data=data: det.forward(data, im_info))
    fl = detector.InFlight(steps)               # the product's own class: every step captured on its own stream
    graphs, streams, outs = fl.graphs, fl.streams, fl.outs

    def run(n, many, sync=False):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n):
            i = (k % NG) if many else 0
            with torch.cuda.stream(streams[i]):
                if H2D:
                    keep[i][1].copy_(hosts[i], non_blocking=True)
                graphs[i].replay()
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        return n * B / (time.perf_counter() - t0)

    n = max(20, 400 // B) // NG * NG
    print('%d graphs round-robin, synchronised after each: %.1f img/s' % (NG, run(2 * NG, True, True)), flush=True)
    print('detections per graph', [int(o['num_detections'].sum().item()) for o in outs], flush=True)
    for _ in range(2):
        print('one graph back to back: %.1f img/s' % run(n, False), flush=True)
        print('%d graphs on %d streams: %.1f img/s' % (NG, NG, run(n, True)), flush=True)
