"""Differentiable (torch-CPU, float64) restatement of the relation module -- TEST INFRASTRUCTURE ONLY: the
checker for the backward kernels.  The forward is oracle/relation.py's (SYM_REL:29-151) line by line; its value is
pinned to that numpy oracle (and through it to the reference's own Python, tests/golden) in
tests/test_oracle_golden.py::test_torch_relation_matches_numpy_oracle.  Gradients come from torch autograd, i.e.
they share no hand-derived formula with the HIP kernels.
"""
import numpy as np
import torch

from . import relation as OR


def relation_module(feat, boxes, p, index=1, nongt_dim=None, group=16):
    """feat: torch [N,1024] float64 (requires_grad allowed); boxes numpy [N,4]; p: dict name -> torch float64
    tensors (requires_grad allowed).  Returns the module output [N,1024]."""
    N = feat.shape[0]
    M = N if nongt_dim is None else nongt_dim
    pe = OR.position_embedding(OR.position_matrix(np.asarray(boxes, np.float32), M))     # [N,M,64] fp32 constants
    E = torch.as_tensor(pe.astype(np.float64))
    g = lambda n: p['%s_%d_%s' % (n.rsplit('_', 1)[0], index, n.rsplit('_', 1)[1])]
    G = torch.relu(E.reshape(N * M, 64) @ g('pair_pos_fc1_weight').t() + g('pair_pos_fc1_bias')).reshape(N, M, group)
    q = feat @ g('query_weight').t() + g('query_bias')
    k = feat[:M] @ g('key_weight').t() + g('key_bias')
    d = q.shape[1] // group
    qh = q.reshape(N, group, d).permute(1, 0, 2)
    kh = k.reshape(M, group, d).permute(1, 0, 2)
    aff = torch.bmm(qh, kh.transpose(1, 2)) * (1.0 / np.sqrt(float(d)))                  # [g,N,M]
    logits = torch.log(torch.clamp(G.permute(2, 0, 1), min=1e-6)) + aff
    S = torch.softmax(logits, dim=2)
    out_t = torch.matmul(S, feat[:M])                                                     # [g,N,1024]
    W = g('linear_out_weight').reshape(group, -1, feat.shape[1])                          # head h -> channels [64h, 64h+64)
    y = torch.einsum('gnf,gcf->ngc', out_t, W).reshape(N, -1) + g('linear_out_bias')
    return y
