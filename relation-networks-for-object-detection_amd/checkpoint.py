"""MXNet `.params` checkpoints for this path: reader / writer of the NDArray-dict binary, the
reference's loader (`lib/utils/load_model.py:12-67`) and its checkpoint callback
(`relation_rcnn/core/callback.py:54-61`), under the same names and argument meaning.

File format (Apache MXNet v1.1.0, `NDArray::Save` list form, src/ndarray/ndarray.cc -- the
library is not vendored in the reference and not installed here, so the layout below is restated
from the published source; PARITY UNPINNED: no `.params` file exists offline to read back):

    uint64  0x112                      kMXAPINDArrayListMagic
    uint64  0                          reserved
    uint64  n                          number of arrays, then n records:
        uint32  0xF993FAC9             NDARRAY_V2_MAGIC   (0xF993FAC8 = V1: no stype field;
                                       anything else = legacy: that word is ndim, dims are uint32)
        int32   stype                  0 = dense (sparse arrays are rejected)
        uint32  ndim,  int64 dim[ndim] TShape
        int32   dev_type, int32 dev_id Context (1 = cpu)
        int32   type_flag              mshadow: 0 f32, 1 f64, 2 f16, 3 u8, 4 i32, 5 i8, 6 i64
        raw little-endian data
    uint64  n                          number of names, then n x (uint64 length, bytes)

Keys are `arg:<name>` / `aux:<name>` (`load_model.py:23-31`).  Arrays are numpy on the host; moving
them to the GPU is the caller's business (Detector / Trainer constructors do it).
"""
import struct

import numpy as np

LIST_MAGIC = 0x112
V1_MAGIC = 0xF993FAC8
V2_MAGIC = 0xF993FAC9
_TYPES = {0: np.float32, 1: np.float64, 2: np.float16, 3: np.uint8, 4: np.int32, 5: np.int8, 6: np.int64}
_FLAGS = {np.dtype(v): k for k, v in _TYPES.items()}


class ParamsFormatError(ValueError):
    pass


def _np(v):
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.ascontiguousarray(v)


def save_ndarray_dict(path, arrays):
    """`mx.nd.save(path, dict)`: arrays = {key: array} (numpy or torch), written in V2 form."""
    names = list(arrays)
    with open(path, 'wb') as f:
        f.write(struct.pack('<QQQ', LIST_MAGIC, 0, len(names)))
        for k in names:
            a = _np(arrays[k])
            if a.dtype not in _FLAGS:
                raise ParamsFormatError("%s: dtype %s has no mshadow type flag" % (k, a.dtype))
            f.write(struct.pack('<Ii', V2_MAGIC, 0))
            f.write(struct.pack('<I', a.ndim))
            f.write(struct.pack('<%dq' % a.ndim, *a.shape))
            f.write(struct.pack('<iii', 1, 0, _FLAGS[a.dtype]))
            f.write(a.astype(a.dtype.newbyteorder('<'), copy=False).tobytes())
        f.write(struct.pack('<Q', len(names)))
        for k in names:
            b = k.encode('utf-8')
            f.write(struct.pack('<Q', len(b)))
            f.write(b)


class _Reader(object):
    def __init__(self, buf):
        self.buf, self.off = buf, 0

    def take(self, fmt):
        n = struct.calcsize(fmt)
        if self.off + n > len(self.buf):
            raise ParamsFormatError("truncated file (need %d bytes at offset %d)" % (n, self.off))
        v = struct.unpack_from(fmt, self.buf, self.off)
        self.off += n
        return v

    def raw(self, n):
        if self.off + n > len(self.buf):
            raise ParamsFormatError("truncated array data (need %d bytes at offset %d)" % (n, self.off))
        v = self.buf[self.off:self.off + n]
        self.off += n
        return v


def _read_array(r):
    (magic,) = r.take('<I')
    if magic == V2_MAGIC:
        (stype,) = r.take('<i')
        if stype != 0:
            raise ParamsFormatError("sparse NDArray (storage type %d) is not supported" % stype)
        (ndim,) = r.take('<I')
        shape = r.take('<%dq' % ndim)
    elif magic == V1_MAGIC:
        (ndim,) = r.take('<I')
        shape = r.take('<%dq' % ndim)
    else:                                   # legacy: the word just read is ndim, uint32 dims
        ndim = magic
        if ndim > 32:
            raise ParamsFormatError("bad NDArray header word 0x%x" % magic)
        shape = r.take('<%dI' % ndim)
    if ndim == 0:                           # is_none(): nothing else is stored
        return None
    _dev_type, _dev_id, flag = r.take('<iii')
    if flag not in _TYPES:
        raise ParamsFormatError("unknown mshadow type flag %d" % flag)
    dt = np.dtype(_TYPES[flag]).newbyteorder('<')
    n = int(np.prod(shape, dtype=np.int64))
    return np.frombuffer(r.raw(n * dt.itemsize), dtype=dt).reshape(shape).astype(_TYPES[flag])


def load_ndarray_dict(path):
    """`mx.nd.load(path)` for the dict form: {key: numpy array}."""
    with open(path, 'rb') as f:
        r = _Reader(f.read())
    magic, _reserved, n = r.take('<QQQ')
    if magic != LIST_MAGIC:
        raise ParamsFormatError("%s is not an MXNet NDArray list (magic 0x%x)" % (path, magic))
    arrays = [_read_array(r) for _ in range(n)]
    (nn,) = r.take('<Q')
    if nn != n:
        raise ParamsFormatError("%d arrays but %d names (a list, not a dict, was saved)" % (n, nn))
    out = {}
    for a in arrays:
        (ln,) = r.take('<Q')
        out[bytes(r.raw(ln)).decode('utf-8')] = a
    return out


# ---------------------------------------------------------------------------------------
# lib/utils/load_model.py
# ---------------------------------------------------------------------------------------
def load_checkpoint(prefix, epoch):
    """(arg_params, aux_params) of '<prefix>-<epoch:04d>.params' (load_model.py:12-31)."""
    save_dict = load_ndarray_dict('%s-%04d.params' % (prefix, epoch))
    arg_params, aux_params = {}, {}
    for k, v in save_dict.items():
        tp, name = k.split(':', 1)
        if tp == 'arg':
            arg_params[name] = v
        if tp == 'aux':
            aux_params[name] = v
    return arg_params, aux_params


def load_param(prefix, epoch, convert=False, ctx=None, process=False):
    """load_model.py:47-67.  `process=True` (test time) moves every `*_test` tensor over its
    training-time name: the de-normalised `bbox_pred_{weight,bias}_test` written by `do_checkpoint`
    become `bbox_pred_{weight,bias}`.  convert / ctx are accepted for signature parity (arrays stay on
    the host; the Detector / Trainer constructors upload them)."""
    arg_params, aux_params = load_checkpoint(prefix, epoch)
    if process:
        for test in [k for k in arg_params if '_test' in k]:
            arg_params[test.replace('_test', '')] = arg_params.pop(test)
    return arg_params, aux_params


def save_checkpoint(prefix, epoch, arg_params, aux_params):
    """The `.params` half of mx.model.save_checkpoint (the symbol json is not part of this path)."""
    d = {'arg:%s' % k: v for k, v in arg_params.items()}
    d.update({'aux:%s' % k: v for k, v in aux_params.items()})
    path = '%s-%04d.params' % (prefix, epoch)
    save_ndarray_dict(path, d)
    return path


def do_checkpoint(prefix, means, stds):
    """core/callback.py:54-61: besides the training weights, store the bbox regression layer with the
    target normalisation folded in (`W_test = (W^T * stds)^T`, `b_test = b * stds + means`, stds / means
    tiled over the regression classes by the caller as in train_end2end.py:149-152)."""
    means = np.asarray(means, np.float32).reshape(-1)
    stds = np.asarray(stds, np.float32).reshape(-1)

    def _callback(iter_no, sym, arg, aux):
        w, b = _np(arg['bbox_pred_weight']).astype(np.float32), _np(arg['bbox_pred_bias']).astype(np.float32)
        arg = dict(arg)
        arg['bbox_pred_weight_test'] = (w.T * stds).T
        arg['bbox_pred_bias_test'] = b * stds + means
        return save_checkpoint(prefix, iter_no + 1, arg, aux)
    return _callback


def merge_params(arg_params, aux_params):
    """One flat {name: torch tensor} dict, the form Detector / Trainer / Backbone take."""
    import torch
    out = {}
    for d in (arg_params, aux_params):
        for k, v in d.items():
            out[k] = torch.as_tensor(np.array(v)) if not hasattr(v, 'detach') else v
    return out
