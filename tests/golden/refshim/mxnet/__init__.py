"""Eager numpy stand-in for the subset of Apache MXNet v1.1.0 that the reference's
hot-path graph code calls.  TEST INFRASTRUCTURE ONLY.

Purpose: the reference (msracver/Relation-Networks-for-Object-Detection) is written
against MXNet 1.1.0, which is not vendored in the reference tree and cannot be
installed here.  `tests/golden/gen_golden.py` puts this directory on sys.path so
that the REFERENCE'S OWN python (symbols/*.py relation module, operator_py/learn_nms.py)
executes unchanged and produces the golden vectors committed under tests/golden/.
What is pinned that way is the reference's op wiring (reshape codes, transposes,
slicing, op order); what is restated here from the MXNet 1.1.0 operator
documentation (and therefore "unpinned") is the semantics of each individual mx
operator listed below.

Numerics: element-wise ops run in float32 exactly as listed (one rounding per op,
like MXNet's fp32 kernels; sin/cos/log/exp are correctly rounded, see `_cr`).  Contractions (FullyConnected, dot, batch_dot,
Convolution 1x1, softmax sums) accumulate in float64 and round once to float32, so
goldens sit at the mathematically exact value of the fp32 graph +-1ulp instead of
depending on a BLAS summation order.

Nothing in the shipped package imports this module.
"""
import sys
import types

import numpy as np

F32 = np.float32

#: parameter registry used by symbolic-style calls that name their weights
#: (mx.sym.FullyConnected(name='query_1', ...) reads PARAMS['query_1_weight']).
PARAMS = {}

#: tensors captured from named ops (e.g. TRACE['softmax_1'] = (input, output)), so a
#: golden run can record intermediates the reference method does not return.
TRACE = {}


def _shape_codes(in_shape, codes):
    """MXNet Reshape special codes 0, -1, -2, -3 (v1.1.0 docs of `Reshape`)."""
    out, i, infer = [], 0, None
    codes = [int(c) for c in codes]
    k = 0
    while k < len(codes):
        c = codes[k]
        if c > 0:
            out.append(c); i += 1
        elif c == 0:
            out.append(in_shape[i]); i += 1
        elif c == -1:
            infer = len(out); out.append(-1); i += 1
        elif c == -2:
            out.extend(in_shape[i:]); i = len(in_shape)
        elif c == -3:
            out.append(in_shape[i] * in_shape[i + 1]); i += 2
        else:
            raise NotImplementedError("reshape code %d" % c)
        k += 1
    if infer is not None:
        known = 1
        for j, v in enumerate(out):
            if j != infer:
                known *= v
        total = int(np.prod(in_shape)) if len(in_shape) else 1
        out[infer] = total // known if known else 0
    return tuple(int(v) for v in out)


class _Ctx(object):
    device_id = 0


class NDArray(object):
    """float32 tensor with the operator overloads the reference uses."""
    __array_priority__ = 100.0

    def __init__(self, a):
        if isinstance(a, NDArray):
            a = a.a
        self.a = np.ascontiguousarray(np.asarray(a, dtype=F32))

    # -- python protocol ----------------------------------------------------
    @property
    def shape(self):
        return tuple(self.a.shape)

    @property
    def context(self):
        return _Ctx()

    def asnumpy(self):
        return self.a.copy()

    def __len__(self):
        return self.a.shape[0]

    def __getitem__(self, k):
        return NDArray(self.a[k])

    def __setitem__(self, k, v):
        self.a[k] = _np(v)

    def _bin(self, o, f, rev=False):
        o = _np(o)
        o = F32(o) if np.isscalar(o) else o
        return NDArray(f(o, self.a) if rev else f(self.a, o))

    def __add__(self, o): return self._bin(o, np.add)
    def __radd__(self, o): return self._bin(o, np.add, True)
    def __sub__(self, o): return self._bin(o, np.subtract)
    def __rsub__(self, o): return self._bin(o, np.subtract, True)
    def __mul__(self, o): return self._bin(o, np.multiply)
    def __rmul__(self, o): return self._bin(o, np.multiply, True)
    def __truediv__(self, o): return self._bin(o, np.divide)
    def __rtruediv__(self, o): return self._bin(o, np.divide, True)
    __div__ = __truediv__
    __rdiv__ = __rtruediv__
    def __neg__(self): return NDArray(-self.a)

    # -- methods ------------------------------------------------------------
    def transpose(self, axes=None):
        return transpose(self, axes=axes)

    def take(self, indices):
        return take(self, indices)

    def max(self, axis=None):
        return NDArray(self.a.max(axis=axis))

    def mean(self, axis=None):
        return NDArray(self.a.astype(np.float64).mean(axis=axis))

    def reshape(self, shape):
        return Reshape(self, shape=shape)


def _np(x):
    return x.a if isinstance(x, NDArray) else x


def _first(args, kw, *names):
    if args:
        return args[0]
    for n in names:
        if n in kw:
            return kw[n]
    raise TypeError("missing tensor argument %s" % (names,))


# ---- creation ---------------------------------------------------------------
def array(a, ctx=None, dtype=None):
    return NDArray(a)


def arange(start, stop=None, step=1.0, **kw):
    if stop is None:
        start, stop = 0, start
    return NDArray(np.arange(float(start), float(stop), float(step)))


def full(shape, val, **kw):
    return NDArray(np.full(tuple(int(s) for s in shape), val, dtype=F32))


def zeros(shape, ctx=None, **kw):
    return NDArray(np.zeros(tuple(int(s) for s in shape), dtype=F32))


def zeros_like(data=None, **kw):
    return NDArray(np.zeros_like(_np(data)))


# ---- shape ops ----------------------------------------------------------------
def Reshape(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    return NDArray(x.reshape(_shape_codes(x.shape, kw['shape'])))


reshape = Reshape


def transpose(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    axes = kw.get('axes', None)
    if axes is None or len(axes) == 0:
        return NDArray(x.T)
    return NDArray(np.transpose(x, tuple(axes)))


def expand_dims(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    axis = a[1] if len(a) > 1 else kw['axis']
    return NDArray(np.expand_dims(x, axis))


def slice_axis(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    axis, begin, end = kw['axis'], kw['begin'], kw['end']
    sl = [slice(None)] * x.ndim
    sl[axis] = slice(begin, end)
    return NDArray(x[tuple(sl)])


def split(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    n, axis = kw['num_outputs'], kw.get('axis', 1)
    squeeze = bool(kw.get('squeeze_axis', 0))
    parts = np.split(x, n, axis=axis)
    if squeeze:
        parts = [np.squeeze(p, axis=axis) for p in parts]
    return [NDArray(p) for p in parts]


def concat(*a, **kw):
    return NDArray(np.concatenate([_np(x) for x in a], axis=kw.get('dim', 1)))


Concat = concat


def tile(*a, **kw):
    return NDArray(np.tile(_np(_first(a, kw, 'data')), kw['reps']))


def reverse(*a, **kw):
    return NDArray(np.flip(_np(_first(a, kw, 'data')), axis=kw['axis']))


def broadcast_to(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    shape = tuple(x.shape[i] if s == 0 else s for i, s in enumerate(kw['shape']))
    return NDArray(np.broadcast_to(x, shape))


def take(a=None, indices=None, **kw):
    x = _np(a)
    idx = np.asarray(_np(indices)).astype(np.int64)
    idx = np.clip(idx, 0, x.shape[0] - 1)          # MXNet default mode='clip'
    return NDArray(x[idx])


def pick(data=None, index=None, axis=-1, **kw):
    x = _np(data)
    idx = np.asarray(_np(index)).astype(np.int64)
    return NDArray(np.take_along_axis(x, np.expand_dims(idx, axis), axis).squeeze(axis))


def BlockGrad(*a, **kw):
    return NDArray(_first(a, kw, 'data'))


# ---- element-wise ---------------------------------------------------------------
def _un(f):
    def op(*a, **kw):
        return NDArray(f(_np(_first(a, kw, 'data'))))
    return op


def _cr(f):
    """Transcendental evaluated in float64 and rounded once to float32 (= a correctly
    rounded fp32 libm).  numpy's own float32 sin/cos/log/exp are SIMD approximations
    that miss correct rounding in 8-40 % of arguments and differ between numpy builds;
    the log -> x100 -> sin chain of the geometry embedding amplifies a 1-ulp log
    difference to ~5e-5 in the embedding, so goldens must not depend on that."""
    return _un(lambda x: f(x.astype(np.float64)))


sin = _cr(np.sin)
cos = _cr(np.cos)
log = _cr(np.log)
exp = _cr(np.exp)
abs = _un(np.abs)


def _bc(f):
    def op(*a, **kw):
        lhs = _np(a[0] if a else kw['lhs'])
        rhs = _np(a[1] if len(a) > 1 else kw['rhs'])
        lhs = F32(lhs) if np.isscalar(lhs) else lhs
        rhs = F32(rhs) if np.isscalar(rhs) else rhs
        return NDArray(f(lhs, rhs))
    return op


broadcast_add = _bc(np.add)
broadcast_minus = _bc(np.subtract)
broadcast_sub = broadcast_minus
broadcast_mul = _bc(np.multiply)
broadcast_div = _bc(np.divide)
broadcast_power = _bc(np.power)
broadcast_maximum = _bc(np.maximum)
broadcast_minimum = _bc(np.minimum)


def maximum(*a, **kw):
    lhs = a[0] if a else kw.get('left', kw.get('lhs'))
    rhs = a[1] if len(a) > 1 else kw.get('right', kw.get('rhs'))
    lhs, rhs = _np(lhs), _np(rhs)
    lhs = F32(lhs) if np.isscalar(lhs) else lhs
    rhs = F32(rhs) if np.isscalar(rhs) else rhs
    return NDArray(np.maximum(lhs, rhs))


def Activation(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    t = kw['act_type']
    if t == 'relu':
        return NDArray(np.maximum(x, F32(0)))
    if t == 'sigmoid':
        return NDArray(F32(1) / (F32(1) + np.exp(-x.astype(np.float64)).astype(F32)))
    raise NotImplementedError(t)


# ---- reductions / sorting ---------------------------------------------------------
def softmax(*a, **kw):
    x = _np(_first(a, kw, 'data')).astype(np.float64)
    axis = kw.get('axis', -1)
    e = np.exp(x - x.max(axis=axis, keepdims=True))
    out = NDArray(e / e.sum(axis=axis, keepdims=True))
    if kw.get('name'):
        TRACE[kw['name']] = (x.astype(F32), out.a.copy())
    return out


def SoftmaxActivation(*a, **kw):
    return softmax(_first(a, kw, 'data'), axis=1)


def smooth_l1(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    s2 = F32(kw.get('scalar', 1.0)) ** 2
    return NDArray(np.where(np.abs(x) < F32(1.0) / s2, F32(0.5) * s2 * x * x, np.abs(x) - F32(0.5) / s2))


def sum(*a, **kw):
    return NDArray(_np(_first(a, kw, 'data')).astype(np.float64).sum(axis=kw.get('axis')))


def sort(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    axis = kw.get('axis', -1)
    s = np.sort(x, axis=axis, kind='stable')
    if not kw.get('is_ascend', True):
        s = np.flip(s, axis=axis)
    return NDArray(s)


def argsort(*a, **kw):
    """Indices as float32 (MXNet's default argsort dtype).  Descending order is
    produced as a stable sort of the negated keys, i.e. ties keep ascending
    index order -- MXNet leaves tie order unspecified; goldens avoid ties."""
    x = _np(_first(a, kw, 'data'))
    axis = kw.get('axis', -1)
    if kw.get('is_ascend', True):
        idx = np.argsort(x, axis=axis, kind='stable')
    else:
        idx = np.argsort(-x, axis=axis, kind='stable')
    return NDArray(idx)


# ---- contractions -------------------------------------------------------------------
def _param(kw, key, name):
    if kw.get(key) is not None:
        return _np(kw[key])
    return _np(PARAMS[name + '_' + key])


def FullyConnected(*a, **kw):
    x = _np(_first(a, kw, 'data'))
    x2 = x.reshape(x.shape[0], -1).astype(np.float64)
    w = _param(kw, 'weight', kw.get('name')).astype(np.float64)
    assert w.shape[0] == int(kw['num_hidden']), (w.shape, kw['num_hidden'])
    y = x2 @ w.T
    if not kw.get('no_bias', False):
        y = y + _param(kw, 'bias', kw.get('name')).astype(np.float64)
    return NDArray(y)


def dot(*a, **kw):
    lhs = _np(a[0] if a else kw['lhs']).astype(np.float64)
    rhs = _np(a[1] if len(a) > 1 else kw['rhs']).astype(np.float64)
    if kw.get('transpose_a'):
        lhs = lhs.T
    if kw.get('transpose_b'):
        rhs = rhs.T
    return NDArray(lhs @ rhs)


def batch_dot(*a, **kw):
    lhs = _np(a[0] if a else kw['lhs']).astype(np.float64)
    rhs = _np(a[1] if len(a) > 1 else kw['rhs']).astype(np.float64)
    if kw.get('transpose_a'):
        lhs = lhs.transpose(0, 2, 1)
    if kw.get('transpose_b'):
        rhs = rhs.transpose(0, 2, 1)
    return NDArray(np.matmul(lhs, rhs))


def Convolution(*a, **kw):
    """Only the 1x1 (optionally grouped) form the relation modules use."""
    x = _np(_first(a, kw, 'data')).astype(np.float64)
    assert tuple(kw['kernel']) == (1, 1)
    w = _param(kw, 'weight', kw.get('name')).astype(np.float64)
    g = int(kw.get('num_group', 1))
    n, c, h, wd = x.shape
    o = int(kw['num_filter'])
    assert w.shape[0] == o and w.shape[1] == c // g, (w.shape, x.shape, g)
    xg = x.reshape(n, g, c // g, h, wd)
    wg = w.reshape(g, o // g, c // g)
    y = np.einsum('ngchw,goc->ngohw', xg, wg).reshape(n, o, h, wd)
    if not kw.get('no_bias', False):
        y = y + _param(kw, 'bias', kw.get('name')).astype(np.float64).reshape(1, o, 1, 1)
    return NDArray(y)


# ---- mx.operator ----------------------------------------------------------------------
class CustomOp(object):
    def assign(self, dst, req, src):
        if req == 'null':
            return
        src = _np(src)
        if req == 'add':
            dst.a[...] = dst.a + src
        else:
            dst.a[...] = src


class CustomOpProp(object):
    def __init__(self, need_top_grad=False):
        self.need_top_grad = need_top_grad


_REGISTRY = {}


def register(name):
    def deco(cls):
        _REGISTRY[name] = cls
        return cls
    return deco


def _module(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_this = sys.modules[__name__]
_tensor_api = {k: v for k, v in list(_this.__dict__.items())
               if callable(v) and not k.startswith('_')}
nd = _module(__name__ + '.nd', **_tensor_api)
ndarray = nd
sym = _module(__name__ + '.sym', Symbol=NDArray, Variable=None, **_tensor_api)
symbol = _module(__name__ + '.symbol', Symbol=NDArray, **_tensor_api)
operator = _module(__name__ + '.operator', CustomOp=CustomOp, CustomOpProp=CustomOpProp,
                   register=register)
random = _module(__name__ + '.random')
contrib = _module(__name__ + '.contrib')
