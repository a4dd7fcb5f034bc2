"""`mx.nd` subset: a thin array handle over a torch tensor (host for parameter initialisation, device for executor
outputs) with the methods the reference's drivers touch: `asnumpy`, `shape`, `context`, `copyto`, `as_in_context`,
arithmetic; constructors `array`, `zeros`, `ones`, `full`, `save` / `load` (the `.params` format of checkpoint.py)."""
import numpy as np
import torch


class Context(object):
    def __init__(self, device_type='gpu', device_id=0):
        self.device_type, self.device_id = device_type, device_id

    def __repr__(self):
        return '%s(%d)' % (self.device_type, self.device_id)

    def __eq__(self, o):
        return isinstance(o, Context) and (self.device_type, self.device_id) == (o.device_type, o.device_id)

    def __hash__(self):
        return hash((self.device_type, self.device_id))


def cpu(device_id=0):
    return Context('cpu', device_id)


def gpu(device_id=0):
    return Context('gpu', device_id)


def _dev(ctx):
    if ctx is None or ctx.device_type == 'cpu':
        return 'cpu'
    return 'cuda:%d' % ctx.device_id


class NDArray(object):
    __array_priority__ = 100.0

    def __init__(self, data):
        self.data = data.data if isinstance(data, NDArray) else data

    @property
    def shape(self):
        return tuple(self.data.shape)

    @property
    def dtype(self):
        return np.dtype(str(self.data.dtype).replace('torch.', '')) if self.data.dtype != torch.bfloat16 else np.float32

    @property
    def context(self):
        d = self.data.device
        return Context('gpu' if d.type == 'cuda' else 'cpu', d.index or 0)

    @property
    def T(self):
        return NDArray(self.data.t())

    def asnumpy(self):
        return self.data.detach().float().cpu().numpy() if self.data.dtype == torch.bfloat16 else self.data.detach().cpu().numpy()

    def asscalar(self):
        return self.asnumpy().reshape(-1)[0]

    def copyto(self, other):
        if isinstance(other, Context):
            return NDArray(self.data.to(_dev(other)).clone())
        other.data.copy_(self.data)
        return other

    def as_in_context(self, ctx):
        return NDArray(self.data.to(_dev(ctx)))

    def astype(self, dtype):
        return NDArray(self.data.to(getattr(torch, np.dtype(dtype).name)))

    def reshape(self, shape):
        return NDArray(self.data.reshape(tuple(shape)))

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, k):
        return NDArray(self.data[k])

    def __setitem__(self, k, v):
        self.data[k] = v.data if isinstance(v, NDArray) else v

    def _b(self, o, f, rev=False):
        o = o.data if isinstance(o, NDArray) else o
        return NDArray(f(o, self.data) if rev else f(self.data, o))

    def __add__(self, o): return self._b(o, torch.add)
    def __radd__(self, o): return self._b(o, torch.add)
    def __sub__(self, o): return self._b(o, torch.sub)
    def __rsub__(self, o): return self._b(o, lambda a, b: a - b, True)
    def __mul__(self, o): return self._b(o, torch.mul)
    def __rmul__(self, o): return self._b(o, torch.mul)
    def __truediv__(self, o): return self._b(o, torch.div)
    __div__ = __truediv__
    def __neg__(self): return NDArray(-self.data)

    def __repr__(self):
        return '<NDArray %s @%s>' % ('x'.join(str(s) for s in self.shape), self.context)


def array(source, ctx=None, dtype=None):
    a = np.asarray(source.asnumpy() if isinstance(source, NDArray) else source, dtype=dtype or np.float32)
    return NDArray(torch.as_tensor(a).to(_dev(ctx)))


def zeros(shape, ctx=None, dtype=None, **kw):
    return NDArray(torch.zeros(tuple(shape) if not isinstance(shape, int) else (shape,), device=_dev(ctx)))


def ones(shape, ctx=None, dtype=None, **kw):
    return NDArray(torch.ones(tuple(shape) if not isinstance(shape, int) else (shape,), device=_dev(ctx)))


def full(shape, val, ctx=None, dtype=None, **kw):
    return NDArray(torch.full(tuple(shape) if not isinstance(shape, int) else (shape,), float(val), device=_dev(ctx)))


def save(fname, data):
    from .. import checkpoint
    checkpoint.save_ndarray_dict(fname, {k: (v.asnumpy() if isinstance(v, NDArray) else v) for k, v in data.items()})


def load(fname):
    from .. import checkpoint
    return {k: array(v) for k, v in checkpoint.load_ndarray_dict(fname).items()}
