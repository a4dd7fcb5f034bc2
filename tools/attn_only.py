"""Launch only the relation kernels (geometry bias, projections, attention) at the bench
configuration -- used under rocprofv3 --pmc to read the attention kernel's HBM traffic."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden'))
import torch
import relnet_amd
from relnet_amd import relation, ops
import cases
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
boxes, feat, p = cases.relation_case(300, 300, 5, 0.01)
pt = {k: torch.as_tensor(v) for k, v in p.items()}
mod = relation.RelationParams(pt, 1, torch.bfloat16, 'cuda')
wp_t, bp = relation.pack_pair_pos([mod], 'cuda')
f = torch.randn(B, 300, 1024, device='cuda').to(torch.bfloat16)
bx = torch.as_tensor(boxes).cuda()[None].repeat(B, 1, 1).contiguous()
bias = ops.geometry_bias(bx, wp_t, bp, 300, half=True)[0]   # fp16 bias -> the LDS throughput kernel
for _ in range(iters):
    relation._module_forward(f, mod, bias, 300, False, True, False)   # as in the pipeline: ReLU(out + resid) only
torch.cuda.synchronize()
print('done')
