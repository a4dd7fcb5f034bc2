"""GPU parity: proposal path (decode -> top-K sort -> NMS -> rois) and ROIPooling, through
the C-ABI, against the CPU oracle.  Bar: proposal indices bit-exact (north_star), box
coordinates bit-exact, ROI pooling bit-exact (it is a max)."""
import ctypes

import numpy as np
import pytest
import torch

import cases
from oracle import proposal as OP
from oracle import nms as ON
from oracle import roi_pooling as ORP

pytestmark = pytest.mark.gpu

CFG = dict(feat_stride=16, scales=(4, 8, 16, 32), ratios=(0.5, 1, 2))


@pytest.fixture(scope='module')
def rn():
    import relnet_amd  # noqa: F401
    from relnet_amd import ops, lib, operator_py
    lib.load()
    return ops, operator_py, lib


def _dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


@pytest.mark.parametrize('seed,delta_sigma,thresh', [(6, 0.5, 0.7), (7, 0.05, 0.7), (8, 0.2, 0.5)])
def test_proposal_bit_exact(rn, seed, delta_sigma, thresh):
    ops, operator_py, _ = rn
    from relnet_amd.operator_py import proposal as P
    cls_prob, deltas, im_info = cases.rpn_case(seed, delta_sigma=delta_sigma)
    rois_o, scores_o, dbg = OP.proposal(cls_prob, deltas, im_info, pre_nms_top_n=6000, post_nms_top_n=300,
                                        threshold=thresh, min_size=0, return_debug=True, **CFG)
    anchors = _dev(P.generate_anchors(16, CFG['ratios'], CFG['scales']))
    rois, scores, d = P.propose_batch(_dev(cls_prob), _dev(deltas), _dev(im_info), anchors, 16, 6000, 300,
                                      thresh, 0, want_debug=True)
    order = d['order'][0].cpu().numpy()
    assert np.array_equal(order, dbg['order'])                       # sort order bit exact
    det = d['det'][0].cpu().numpy()
    assert np.array_equal(det, dbg['det'])                           # decoded fp32 boxes + scores
    nk = int(d['num_keep'][0])
    keep = d['keep'][0].cpu().numpy()[:min(nk, 300)]
    assert np.array_equal(keep, dbg['keep'][:len(keep)])             # NMS keep set bit exact
    if dbg['n_kept'] >= 300:
        assert np.array_equal(rois[0].cpu().numpy(), rois_o)
        assert np.array_equal(scores[0].cpu().numpy().reshape(-1, 1), scores_o)
    else:                                                            # padded rows: any kept box
        assert nk == dbg['n_kept']
        r = rois[0].cpu().numpy()
        assert np.array_equal(r[:nk], rois_o[:nk])
        kept = {tuple(x) for x in r[:nk]}
        assert all(tuple(x) in kept for x in r[nk:])


def test_proposal_operator_protocol(rn):
    """Same call as the reference graph: mx.sym.Custom(op_type='proposal', ...) with string attrs."""
    ops, operator_py, _ = rn
    cls_prob, deltas, im_info = cases.rpn_case(9)
    want, _ = OP.proposal(cls_prob, deltas, im_info, **CFG)
    rois = operator_py.Custom(cls_prob=_dev(cls_prob), bbox_pred=_dev(deltas), im_info=_dev(im_info),
                              name='rois', op_type='proposal', feat_stride=16, scales=(4, 8, 16, 32),
                              ratios=(0.5, 1, 2), rpn_pre_nms_top_n=6000, rpn_post_nms_top_n=300,
                              threshold=0.7, rpn_min_size=0)
    assert rois.shape == (300, 5) and np.array_equal(rois.cpu().numpy(), want)
    with pytest.raises(ValueError):
        two = _dev(np.concatenate([cls_prob, cls_prob]))
        operator_py.Custom(cls_prob=two, bbox_pred=_dev(np.concatenate([deltas, deltas])),
                           im_info=_dev(np.concatenate([im_info, im_info])), op_type='proposal')


def test_proposal_batch_of_images(rn):
    ops, _, _ = rn
    from relnet_amd.operator_py import proposal as P
    cs = [cases.rpn_case(s) for s in (11, 12, 13)]
    anchors = _dev(P.generate_anchors(16, CFG['ratios'], CFG['scales']))
    rois, scores = P.propose_batch(_dev(np.concatenate([c[0] for c in cs])), _dev(np.concatenate([c[1] for c in cs])),
                                   _dev(np.concatenate([c[2] for c in cs])), anchors, 16, 6000, 300, 0.7, 0)
    for i, c in enumerate(cs):
        want, _ = OP.proposal(*c, **CFG)
        got = rois[i].cpu().numpy()
        assert np.array_equal(got[:, 1:], want[:, 1:]) and (got[:, 0] == i).all()


def test_min_size_filter_and_small_topn(rn):
    ops, _, _ = rn
    from relnet_amd.operator_py import proposal as P
    cls_prob, deltas, im_info = cases.rpn_case(14, delta_sigma=1.0)
    rois_o, _, dbg = OP.proposal(cls_prob, deltas, im_info, pre_nms_top_n=1000, post_nms_top_n=50,
                                 threshold=0.7, min_size=16, return_debug=True, **CFG)
    anchors = _dev(P.generate_anchors(16, CFG['ratios'], CFG['scales']))
    rois, _, d = P.propose_batch(_dev(cls_prob), _dev(deltas), _dev(im_info), anchors, 16, 1000, 50, 0.7, 16,
                                 want_debug=True)
    assert np.array_equal(d['order'][0].cpu().numpy(), dbg['order'])
    assert np.array_equal(rois[0].cpu().numpy(), rois_o)


@pytest.mark.parametrize('n,seed,thresh', [(300, 31, 0.5), (1000, 32, 0.7), (6000, 33, 0.7), (63, 34, 0.3), (1, 35, 0.5)])
def test_reference_nms_c_abi(rn, n, seed, thresh):
    """`_nms` with the reference's prototype (lib/nms/gpu_nms.hpp:1-2): host pointers."""
    _, _, lib = rn
    dets = cases.dets_case(n, seed)
    order = ON.argsort_desc(dets[:, 4])
    sorted_dets = np.ascontiguousarray(dets[order])
    keep = np.zeros(n, dtype=np.int32)
    num = ctypes.c_int(0)
    lib.load()._nms(keep.ctypes.data, ctypes.addressof(num), sorted_dets.ctypes.data, n, 5, thresh, 0)
    assert num.value >= 1
    got = list(order[keep[:num.value]])
    assert got == ON.gpu_nms(dets, thresh)


def _roi_cases(seed, R, B):
    rng = np.random.default_rng(seed)
    boxes = cases.random_boxes(R, seed + 1)
    # edge cases: degenerate, out-of-image, full-image, sub-pixel
    boxes[0] = [0, 0, 0, 0]
    boxes[1] = [990, 590, 999, 599]
    boxes[2] = [0, 0, 999, 599]
    boxes[3] = [500.4, 300.6, 500.5, 300.7]
    boxes[4] = [1200, 700, 1300, 800]            # beyond the map -> empty bins -> 0
    bidx = rng.integers(0, B, R).astype(np.float32)
    return np.hstack((bidx[:, None], boxes)).astype(np.float32)


def test_roi_pooling_nchw_fp32_bit_exact(rn):
    ops, _, _ = rn
    rng = np.random.default_rng(3)
    data = rng.normal(0, 1, (2, 16, 38, 63)).astype(np.float32)
    rois = _roi_cases(4, 40, 2)
    want, warg = ORP.roi_pooling(data, rois, return_argmax=True)
    out, arg = ops.roi_pool(_dev(data), _dev(rois), want_argmax=True)
    assert np.array_equal(out.cpu().numpy(), want)
    assert np.array_equal(arg.cpu().numpy(), warg)


def test_roi_pooling_channels_last_bf16(rn):
    ops, _, _ = rn
    rng = np.random.default_rng(5)
    data = torch.as_tensor(rng.normal(0, 1, (2, 256, 38, 63)).astype(np.float32)).cuda().to(torch.bfloat16)
    rois = _roi_cases(6, 60, 2)
    want = ORP.roi_pooling(data.float().cpu().numpy(), rois)
    cl = data.contiguous(memory_format=torch.channels_last)
    out = ops.roi_pool(cl, _dev(rois), channels_last_out=True)
    assert out.permute(0, 2, 3, 1).is_contiguous()
    assert np.array_equal(out.float().cpu().numpy(), want)


@pytest.mark.parametrize('n,seed,thresh,post', [(6000, 51, 0.7, 300), (1000, 52, 0.3, 100), (700, 53, 0.5, 2048)])
def test_fused_greedy_nms_equals_mask_scan(rn, n, seed, thresh, post):
    ops, _, _ = rn
    dets = cases.dets_case(n, seed)
    order = ON.argsort_desc(dets[:, 4])
    det = _dev(np.stack([dets[order], dets[order][::-1][np.argsort(-dets[order][::-1][:, 4], kind='stable')]]))
    a = ops.nms_sorted(det, thresh, post=post, want_keep=True)
    b = ops.nms_greedy(det, thresh, post, want_keep=True)
    assert torch.equal(a['num_keep'].clamp(max=post), b['num_keep'])
    assert torch.equal(a['rois'], b['rois']) and torch.equal(a['scores'], b['scores'])
    k = int(b['num_keep'][0])
    assert torch.equal(a['keep'][:, :k], b['keep'][:, :k])
    want = ON.nms_sorted_f32(dets[order][:, :4], thresh, max_keep=post)
    assert np.array_equal(b['keep'][0, :len(want)].cpu().numpy(), want)
