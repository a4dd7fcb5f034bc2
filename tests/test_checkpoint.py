"""`.params` checkpoints: the MXNet NDArray-dict binary, `load_param`'s `_test` key swap
(lib/utils/load_model.py:47-67) and `do_checkpoint`'s de-normalised bbox weights (core/callback.py:54-61).
Format parity is UNPINNED (no MXNet and no .params file offline): what is tested is that the writer emits exactly
the byte layout documented from the MXNet v1.1.0 source, that the reader accepts the V2 / V1 / legacy record
forms, and the reference's key handling."""
import struct

import numpy as np
import pytest

import relnet_amd  # noqa: F401
from relnet_amd import checkpoint as ck


def _arrays():
    rng = np.random.default_rng(0)
    return {'arg:conv1_weight': rng.normal(size=(64, 3, 7, 7)).astype(np.float32),
            'arg:bbox_pred_bias': rng.normal(size=(8,)).astype(np.float32),
            'aux:bn_conv1_moving_var': rng.random(64).astype(np.float32),
            'arg:half': rng.normal(size=(3, 5)).astype(np.float16),
            'arg:idx': np.arange(7, dtype=np.int32)}


def test_ndarray_dict_round_trip_and_layout(tmp_path):
    a = _arrays()
    path = str(tmp_path / 'x.params')
    ck.save_ndarray_dict(path, a)
    raw = open(path, 'rb').read()
    assert struct.unpack_from('<QQQ', raw) == (0x112, 0, len(a))                       # list magic, reserved, count
    assert struct.unpack_from('<IiI', raw, 24) == (0xF993FAC9, 0, 4)                    # V2 magic, dense, ndim
    assert struct.unpack_from('<4q', raw, 36) == (64, 3, 7, 7)                          # int64 dims
    assert struct.unpack_from('<iii', raw, 68) == (1, 0, 0)                             # cpu(0), float32
    assert np.array_equal(np.frombuffer(raw, '<f4', 64 * 147, 80).reshape(64, 3, 7, 7), a['arg:conv1_weight'])
    b = ck.load_ndarray_dict(path)
    assert list(b) == list(a)
    for k in a:
        assert b[k].dtype == a[k].dtype and np.array_equal(b[k], a[k]), k


def test_reader_accepts_v1_and_legacy_records(tmp_path):
    x = np.arange(6, dtype=np.float32).reshape(2, 3)
    v1 = struct.pack('<I', 0xF993FAC8) + struct.pack('<I2q', 2, 2, 3) + struct.pack('<iii', 2, 1, 0) + x.tobytes()
    legacy = struct.pack('<I2I', 2, 2, 3) + struct.pack('<iii', 1, 0, 0) + x.tobytes()
    names = struct.pack('<Q', 2) + b''.join(struct.pack('<Q', len(n)) + n for n in (b'arg:a', b'aux:b'))
    path = str(tmp_path / 'old.params')
    open(path, 'wb').write(struct.pack('<QQQ', 0x112, 0, 2) + v1 + legacy + names)
    d = ck.load_ndarray_dict(path)
    assert np.array_equal(d['arg:a'], x) and np.array_equal(d['aux:b'], x)
    open(path, 'wb').write(b'\x00' * 24)
    with pytest.raises(ck.ParamsFormatError):
        ck.load_ndarray_dict(path)
    open(path, 'wb').write(struct.pack('<QQQ', 0x112, 0, 1) + v1[:20])
    with pytest.raises(ck.ParamsFormatError):
        ck.load_ndarray_dict(path)


def test_do_checkpoint_and_load_param_test_swap(tmp_path):
    rng = np.random.default_rng(1)
    arg = {'bbox_pred_weight': rng.normal(size=(8, 1024)).astype(np.float32),
           'bbox_pred_bias': rng.normal(size=(8,)).astype(np.float32),
           'fc_new_2_weight': rng.normal(size=(4, 4)).astype(np.float32)}
    aux = {'bn_conv1_moving_mean': rng.normal(size=(64,)).astype(np.float32)}
    means, stds = np.tile([0.0, 0.0, 0.0, 0.0], 2), np.tile([0.1, 0.1, 0.2, 0.2], 2)
    prefix = str(tmp_path / 'rcnn')
    path = ck.do_checkpoint(prefix, means, stds)(3, None, arg, aux)
    assert path.endswith('rcnn-0004.params') and '_test' not in ''.join(arg)          # caller's dict is left alone
    a1, x1 = ck.load_param(prefix, 4)                                                   # training-time view
    assert set(a1) == set(arg) | {'bbox_pred_weight_test', 'bbox_pred_bias_test'} and set(x1) == set(aux)
    assert np.array_equal(a1['bbox_pred_weight'], arg['bbox_pred_weight'])
    a2, _ = ck.load_param(prefix, 4, process=True)                                      # test-time view: swapped in
    assert set(a2) == set(arg)
    assert np.allclose(a2['bbox_pred_weight'], arg['bbox_pred_weight'] * stds[:, None].astype(np.float32))
    assert np.allclose(a2['bbox_pred_bias'], arg['bbox_pred_bias'] * stds + means)
    assert np.array_equal(a2['fc_new_2_weight'], arg['fc_new_2_weight'])
