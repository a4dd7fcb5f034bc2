import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import relnet_amd
from relnet_amd import backbone
from oracle import network as ON
p = backbone.init_params(seed=1)
g = torch.Generator().manual_seed(0)
data = torch.randn(1, 3, 600, 1000, generator=g)
for th in [int(x) for x in sys.argv[1:]]:
    torch.set_num_threads(th)
    with torch.no_grad():
        t0 = time.time(); c4, c5 = ON.backbone(data, p); t1 = time.time()
        cls, box, feat = ON.rpn_and_feat(c4, c5, p); t2 = time.time()
    print('threads', th, 'backbone %.2fs rpn %.2fs' % (t1 - t0, t2 - t1), flush=True)
t0 = time.time()
out = ON.detect(data, np.array([[600, 1000, 1.0]], np.float32), p)
print('full detect %.2fs' % (time.time() - t0))
