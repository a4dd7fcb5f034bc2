"""Time the fused geometry + attention kernel alone at the bench configuration (env RELNET_FUSED_ABLATE for experiments)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import torch
import relnet_amd
from relnet_amd import relation, ops
import cases
B = int(sys.argv[1]) if len(sys.argv) > 1 else 54
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
boxes, feat, p = cases.relation_case(300, 300, 5, 0.01)
pt = {k: torch.as_tensor(v) for k, v in p.items()}
mod = relation.RelationParams(pt, 1, torch.bfloat16, 'cuda')
f = torch.randn(B, 300, 1024, device='cuda').to(torch.bfloat16)
bx = torch.as_tensor(boxes).cuda()[None].repeat(B, 1, 1).contiguous()
qk = ops.gemm_nt(f.reshape(B * 300, 1024), mod.wqk, mod.bqk).reshape(B, 300, -1)
vwt = torch.zeros((B, 1024, 320), device='cuda', dtype=torch.bfloat16)
ops.gemm_nt(mod.wout, f, out=vwt, n_cols=300)
def run():
    return ops.relation_attention_fused(qk[:, :, :1024], qk[:, :300, 1024:], vwt, bx, mod.wp_dev, mod.bp_dev, bout=mod.bout,
                                        resid=f, M=300, want_out=False, want_act=True)
for _ in range(3):
    run()
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()                   # replay 20 launches per graph: python / ctypes overhead out of the timing
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    run()
    with torch.cuda.graph(g, stream=st):
        for _ in range(20):
            run()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    g.replay()
e1.record(); torch.cuda.synchronize()
print('ablate=%s B=%d: %.1f us / launch' % (os.environ.get('RELNET_FUSED_ABLATE', '0'), B, e0.elapsed_time(e1) * 1e3 / iters / 20))
