"""Which Python line launches which library (at::native / rocclr copy / fill) kernel in one EAGER training step:
    python tools/glue_trace.py [B] [--infer]
Runs two warm steps, then one step under torch.profiler(with_stack=True) and prints, per source line of the package, the
non-relnet device kernels it launched (count, total us).  Round 6: the list behind the glue removals of train.py."""
import collections
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import relnet_amd  # noqa: E402,F401
from relnet_amd import backbone, train, detector  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 1
infer = '--infer' in sys.argv
H, W, G = 600, 1000, 8
params = backbone.init_params(seed=1)
g = torch.Generator().manual_seed(1000)
data = torch.randn(B, 3, H, W, generator=g).cuda()
im_info = torch.tensor([[float(H), float(W), 1.0]] * B).cuda()
if infer:
    det = detector.Detector(params, cfg=detector.Config())
    step = lambda: det.forward(data, im_info)
else:
    cfg = train.TrainConfig.from_experiment('rcnn_end2end_relation_learn_nms_8epoch', train=True)
    tr = train.Trainer(params, cfg, im_hw=(H, W))
    rng = np.random.default_rng(2)
    gt = np.zeros((B, G, 5), np.float32)
    for b in range(B):
        bw, bh = rng.uniform(32, 400, G), rng.uniform(32, 400, G)
        x1, y1 = rng.uniform(0, W - 1 - bw), rng.uniform(0, H - 1 - bh)
        gt[b] = np.stack([x1, y1, x1 + bw, y1 + bh, rng.integers(1, 81, G)], 1)
    gt = torch.as_tensor(gt).cuda()

    def step():
        tr.forward_backward(data, im_info, gt)
        tr.all_reduce(wait=False)
        tr.update()
with torch.no_grad():
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True,
                 experimental_config=torch._C._profiler._ExperimentalConfig(verbose=True)) as prof:
        step()
        torch.cuda.synchronize()
here = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + '/'
by_line = collections.defaultdict(lambda: [0, 0.0, collections.Counter()])
n_all = n_lib = 0
for e in prof.events():
    if e.device_type != torch.autograd.DeviceType.CPU or not e.kernels:
        continue
    for k in e.kernels:
        n_all += 1
        if 'relnet::' in k.name:
            continue
        n_lib += 1
        st = [f for f in (e.stack or []) if 'relation-networks-for-object-detection_amd' in f or 'relnet_amd' in f]
        line = (st[0] if st else (e.stack[0] if e.stack else '?')).replace(here, '')
        r = by_line[line]
        r[0] += 1; r[1] += k.duration; r[2][e.name] += 1
rows = sorted(by_line.items(), key=lambda kv: -kv[1][1])
print('eager %s step at %d image(s): %d device kernels launched from aten / library ops, %d of them not relnet kernels' % ('inference' if infer else 'training', B, n_all, n_lib))
for k, (cnt, t, ops_) in rows:
    print('%4d launches %8.1f us  %s   [%s]' % (cnt, t, k, ', '.join('%s x%d' % kv for kv in ops_.most_common(6))))
