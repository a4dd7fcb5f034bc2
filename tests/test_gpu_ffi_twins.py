"""Python-level FFI twins of lib/nms and lib/bbox against the oracle restatements, which are
pinned to the reference's own numpy `nms` / `soft_nms` / `bbox_overlaps_py` outputs by tests/test_oracle_golden.py."""
import numpy as np
import pytest

import cases
from oracle import nms as ONMS
from oracle import boxes as OB

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def twins():
    import relnet_amd  # noqa: F401
    from relnet_amd import nms, bbox
    return nms, bbox


def _dets(n, seed, dtype):
    rng = np.random.default_rng(seed)
    b = cases.random_boxes(n, seed)
    s = rng.permutation(n).astype(np.float64) / n * 0.98 + 0.01            # tie-free
    return np.hstack((b, s[:, None])).astype(dtype)


@pytest.mark.parametrize('n,thresh', [(1, 0.5), (70, 0.3), (300, 0.5), (1000, 0.7)])
def test_gpu_nms_and_wrappers(twins, n, thresh):
    nms, _ = twins
    d = _dets(n, 5 + n, np.float32)
    want = ONMS.gpu_nms(d.copy(), thresh)                                 # restated nms_kernel.cu (float32 IoU, `>`)
    got = nms.gpu_nms(d, thresh, 0)
    assert [int(i) for i in got] == [int(i) for i in want]
    assert nms.gpu_nms_wrapper(thresh, 0)(d) == got
    assert nms.gpu_nms(np.zeros((0, 5), np.float32), thresh) == []
    # cpu_nms: `>=` suppresses -- identical on tie-free IoUs, and different exactly at ovr == thresh
    assert [int(i) for i in nms.cpu_nms_wrapper(thresh)(d)] == [int(i) for i in want]
    pair = np.array([[0, 0, 9, 9, 0.9], [0, 0, 9, 19, 0.8]], np.float32)  # IoU = 100 / 200 = 0.5 exactly
    assert nms.gpu_nms(pair, 0.5) == [0, 1] and nms.cpu_nms(pair, 0.5) == [0]


@pytest.mark.parametrize('n', [3, 120, 300, 700])
def test_py_nms_and_softnms_wrappers_float64(twins, n):
    nms, _ = twins
    d = _dets(n, 11 + n, np.float64)
    for t in (0.3, 0.5):
        want = ONMS.py_nms(d.copy(), t)
        got = nms.py_nms_wrapper(t)(d.copy())
        assert [int(i) for i in got] == [int(i) for i in want]
    for max_dets in (-1, 100):
        a, b = d.copy(), d.copy()
        want = ONMS.soft_nms(a, 0.6, max_dets)
        got = nms.py_softnms_wrapper(0.6, max_dets)(b)
        assert got.shape == want.shape
        np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-300)
        if max_dets == -1:                                                 # scores written back like nms.py:114
            np.testing.assert_allclose(np.sort(b[:, 4]), np.sort(want[:, 4]), rtol=1e-10, atol=1e-300)
    assert nms.soft_nms(np.zeros((0, 5)), 0.6, -1).shape == (0, 5) and nms.nms(np.zeros((0, 5)), 0.5) == []
    # lib/nms/nms.py takes ANY float64 score column (raw logits, scores <= -1): nothing may be mistaken for a removed slot
    dn = d.copy()
    dn[:, 4] = np.random.default_rng(n).normal(-2.0, 3.0, n)
    want = ONMS.py_nms(dn.copy(), 0.5)
    got = nms.py_nms_wrapper(0.5)(dn.copy())
    assert [int(i) for i in got] == [int(i) for i in want] and (dn[want, 4] < -1.0).any()


def test_bbox_overlaps_cython(twins):
    import torch
    _, bbox = twins
    for (n, k, seed) in ((1, 1, 0), (308, 8, 1), (1000, 37, 2)):
        b = cases.random_boxes(n, 40 + seed).astype(np.float64)
        q = cases.random_boxes(k, 50 + seed).astype(np.float64)
        q[0] = b[0]                                                      # IoU exactly 1
        want = OB.bbox_overlaps(b, q)
        got = bbox.bbox_overlaps_cython(b, q)
        assert got.dtype == np.float64 and got.shape == (n, k)
        assert np.array_equal(got, want) and got[0, 0] == 1.0
        dev = bbox.bbox_overlaps(torch.as_tensor(b).cuda(), torch.as_tensor(q).cuda())
        assert dev.is_cuda and np.array_equal(dev.cpu().numpy(), want)
    assert bbox.bbox_overlaps_cython(np.zeros((0, 4)), np.zeros((3, 4))).shape == (0, 3)
