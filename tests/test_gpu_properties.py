"""Size-independent properties of the HIP path at BASELINE sizes (N = M = 300, 6000 pre-NMS boxes,
600x1000 geometry) and the edge cases of the boundary (empty / ragged / maximum inputs)."""
import ctypes

import numpy as np
import pytest
import torch

import cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def rn():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, relation, lib
    lib.load()
    return ops, relation, lib


def _d(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _module(relation, feat, boxes, pt, dtype):
    return relation.attention_module_multi_head(feat, boxes, pt, dtype=dtype)


@pytest.mark.parametrize('dtype', [torch.float32, torch.bfloat16])
def test_relation_is_permutation_equivariant(rn, dtype):
    """Re-ordering the rois (queries AND keys) re-orders the outputs and changes nothing else."""
    ops, relation, _ = rn
    boxes, feat, p = cases.relation_case(300, 300, 91, 0.02)
    pt = {k: torch.as_tensor(v) for k, v in p.items()}
    perm = np.random.default_rng(1).permutation(300)
    y = _module(relation, _d(feat), _d(boxes), pt, dtype).float()
    yp = _module(relation, _d(feat[perm]), _d(boxes[perm]), pt, dtype).float()
    tol = 2e-5 if dtype == torch.float32 else 2e-2
    assert (yp - y[torch.as_tensor(perm).cuda()]).abs().max().item() <= tol * y.abs().max().item()


def test_attention_rows_are_convex_combinations(rn):
    """softmax weights sum to 1: with constant value rows the output is that constant (+ bias)."""
    ops, relation, _ = rn
    B, N, H = 2, 300, 16
    Mpad = ops.pad32(N)
    g = torch.Generator().manual_seed(3)
    q = torch.randn(B, N, 1024, generator=g).cuda().to(torch.bfloat16)
    k = torch.randn(B, N, 1024, generator=g).cuda().to(torch.bfloat16)
    bias = (torch.randn(B, H, N, Mpad, generator=g) * 2).cuda().to(torch.float16)
    vwt = torch.zeros(B, 1024, Mpad, device='cuda', dtype=torch.bfloat16)
    const = torch.randn(1024, generator=g).cuda().to(torch.bfloat16)
    vwt[:, :, :N] = const[None, :, None]
    bout = torch.randn(1024, generator=g).cuda()
    out, _, _ = ops.relation_attention(q, k, vwt, bias, bout=bout, M=N)
    want = (const.float() + bout)[None, None, :].expand(B, N, 1024)
    assert (out.float() - want).abs().max().item() <= 2e-2 * want.abs().max().item()


def test_nms_is_idempotent_and_sorted_at_full_size(rn):
    """NMS of the kept set keeps everything; kept indices ascend; no kept pair overlaps > thresh."""
    ops, _, _ = rn
    from oracle import nms as ON
    dets = cases.dets_case(6000, 7)
    order = ON.argsort_desc(dets[:, 4])
    det = _d(dets[order][None])
    r = ops.nms_sorted(det, 0.7, want_keep=True, max_keep=6000)
    nk = int(r['num_keep'][0])
    keep = r['keep'][0, :nk].cpu().numpy()
    assert (np.diff(keep) > 0).all()
    kept = dets[order][keep]
    r2 = ops.nms_sorted(_d(kept[None]), 0.7, want_keep=True, max_keep=nk)
    assert int(r2['num_keep'][0]) == nk
    for i in range(0, nk, 97):
        iou = ON.iou_f32(kept[i, :4], kept[:, :4])
        iou[i] = 0
        assert (iou <= np.float32(0.7)).all()


def test_topk_sort_is_sorted_and_a_permutation(rn):
    ops, _, _ = rn
    rng = np.random.default_rng(5)
    scores = rng.random((3, 27528)).astype(np.float32)
    boxes = rng.random((3, 27528, 4)).astype(np.float32)
    det, index, count = ops.topk_sort(_d(scores), _d(boxes), 6000)
    s = det[:, :, 4].cpu().numpy()
    assert (np.diff(s, axis=1) <= 0).all() and (count.cpu().numpy() == 6000).all()
    for b in range(3):
        idx = index[b].cpu().numpy()
        assert len(np.unique(idx)) == 6000
        assert np.array_equal(s[b], scores[b][idx]) and s[b].min() >= np.sort(scores[b])[-6000]


def test_empty_and_degenerate_inputs(rn):
    ops, relation, lib = rn
    # `_nms` with zero boxes (reference: gpu_nms on an empty array never reaches _nms; here it is defined)
    num = ctypes.c_int(-5)
    lib.load()._nms(None, ctypes.addressof(num), None, 0, 5, 0.5, 0)
    assert num.value == 0
    # a single roi attends only to itself: the module output is V Wout^T + bias of that roi
    boxes, feat, p = cases.relation_case(1, 1, 93, 0.02)
    pt = {k: torch.as_tensor(v) for k, v in p.items()}
    y = relation.attention_module_multi_head(_d(feat), _d(boxes), pt, dtype=torch.float32).cpu().numpy()
    wo = p['linear_out_1_weight'].reshape(16, 64, 1024)
    want = np.einsum('hoc,c->ho', wo.astype(np.float64), feat[0].astype(np.float64)).reshape(1024) + p['linear_out_1_bias']
    assert np.abs(y[0] - want).max() <= 1e-4 * np.abs(want).max()
    # all-identical boxes: NMS keeps exactly one
    same = np.tile(np.array([[10, 10, 50, 60, 0.0]], np.float32), (130, 1))
    same[:, 4] = np.linspace(1, 0.1, 130)
    r = ops.nms_greedy(_d(same[None]), 0.5, 300)
    assert int(r['num_keep'][0]) == 1 and torch.equal(r['rois'][0, 0, 1:], _d(same[0, :4]))
    assert torch.equal(r['rois'][0, 299], r['rois'][0, 0])              # padded with the kept box
    # ragged batch: per-image counts limit the scan
    dets = cases.dets_case(256, 9)
    from oracle import nms as ON
    order = ON.argsort_desc(dets[:, 4])
    det = _d(np.stack([dets[order], dets[order]]))
    cnt = torch.tensor([256, 100], dtype=torch.int32).cuda()
    r = ops.nms_greedy(det, 0.5, 300, counts=cnt, want_keep=True)
    want1 = ON.nms_sorted_f32(dets[order][:100, :4], 0.5)
    assert int(r['num_keep'][1]) == len(want1)
    assert np.array_equal(r['keep'][1, :len(want1)].cpu().numpy(), want1)


def test_gemm_is_linear_at_fc1_size(rn):
    """fc_new_1 shape (K = 12544): f(a + b) = f(a) + f(b) in fp32 up to rounding."""
    ops, _, _ = rn
    g = torch.Generator().manual_seed(11)
    a = torch.randn(600, 12544, generator=g).cuda()
    b = torch.randn(600, 12544, generator=g).cuda()
    w = (torch.randn(1024, 12544, generator=g) * 0.01).cuda()
    fa, fb, fab = ops.gemm_nt(a, w), ops.gemm_nt(b, w), ops.gemm_nt(a + b, w)
    assert (fab - (fa + fb)).abs().max().item() <= 2e-5 * fab.abs().max().item()
