"""`mx`: the slice of the MXNet v1.1.0 Python API that the reference's graph files and operators are written
against (graph-builder row), backed by librelnet_hip.so.

    from relnet_amd import mx             # or: relnet_amd.mx.install(); import mxnet as mx
    sym = SomeReferenceSymbolClass().get_symbol(cfg, is_train=False)      # relation_rcnn/symbols/*.py, unchanged
    exe = sym.bind(mx.gpu(0), args=arg_params, aux_states=aux_params)     # torch / numpy / NDArray values
    rois, cls_prob, bbox_pred = exe.forward(is_train=False, data=..., im_info=...)[:3]

`install()` registers this package as `mxnet` in sys.modules together with the sibling twins the reference's
files import by their repository-relative names (`operator_py.*`, `utils.symbol`, `nms.nms`, `bbox.bbox_transform`)
and, for the Python-2 sources, the `cPickle` / `xrange` / `np.float` aliases -- nothing in a reference file is edited.
"""
import sys
import types

from . import registry as _R
from . import graph as _S
from . import ndarray as nd
from .ndarray import Context, cpu, gpu, NDArray  # noqa: F401
from .graph import Symbol  # noqa: F401


def _make_namespace(name):
    m = types.ModuleType(name)

    def ctor(op):
        def f(*args, **kwargs):
            return _R.make(op, args, kwargs)
        f.__name__ = op
        return f
    for op in _R.OPS:
        if not op.startswith('_') and op != 'Custom':
            setattr(m, op, ctor(op))
    m.Variable, m.var, m.Group, m.Symbol = _S.Variable, _S.var, _S.Group, _S.Symbol
    m.load, m.load_json, m.reset_names = _S.load, _S.load_json, _S.reset_names
    m.Custom = lambda *a, **k: _R.make_custom(a, k)

    def _two(sym_op, scalar_op, rscalar_op=None):
        def f(left=None, right=None, **kw):
            left = kw.pop('lhs', left); right = kw.pop('rhs', right)
            ls, rs = isinstance(left, _S.Symbol), isinstance(right, _S.Symbol)
            if ls and rs:
                return _R.make(sym_op, [left, right], kw)
            if ls:
                return _R.make(scalar_op, [left], dict(kw, scalar=float(right)))
            if rs:
                return _R.make(rscalar_op or scalar_op, [right], dict(kw, scalar=float(left)))
            raise TypeError("at least one Symbol expected")
        return f
    m.maximum = _two('_maximum', '_maximum_scalar')
    m.minimum = _two('_minimum', '_minimum_scalar')
    m.pow = m.power = _two('_power', '_power_scalar', '_rpower_scalar')      # pow(2, sym) = 2 ** sym, not sym ** 2
    return m


sym = symbol = _make_namespace(__name__ + '.symbol_api')

contrib = types.ModuleType(__name__ + '.contrib')
contrib.sym = contrib.symbol = types.ModuleType(__name__ + '.contrib.symbol')
for _pub, _op in (('DeformableConvolution', '_contrib_DeformableConvolution'),
                  ('DeformablePSROIPooling', '_contrib_DeformablePSROIPooling'), ('Proposal', '_contrib_Proposal'),
                  ('ROIAlign', '_contrib_ROIAlign')):
    setattr(contrib.sym, _pub, (lambda o: lambda *a, **k: _R.make(o, a, k))(_op))

# mx.operator: the CustomOp protocol lives in relnet_amd.operator_py (same classes, same registry)
operator = types.ModuleType(__name__ + '.operator')


def _bind_operator_namespace():
    from .. import operator_py
    operator.CustomOp, operator.CustomOpProp, operator.register = operator_py.CustomOp, operator_py.CustomOpProp, operator_py.register


random = types.ModuleType(__name__ + '.random')
_gen = None


def _seed(s):
    global _gen
    import torch
    _gen = torch.Generator().manual_seed(int(s))


def _normal(loc=0.0, scale=1.0, shape=(1,), ctx=None, **kw):
    import torch
    if _gen is None:
        _seed(0)
    return NDArray(torch.randn(tuple(shape) if not isinstance(shape, int) else (shape,), generator=_gen) * scale + loc)


random.seed, random.normal = _seed, _normal


def install(py2_shims=True):
    """Make `import mxnet` (and the reference's sibling imports) resolve to this package."""
    import importlib
    _bind_operator_namespace()
    me = sys.modules[__name__]
    sys.modules['mxnet'] = me
    sys.modules['mxnet.symbol'] = sym
    sys.modules['mxnet.ndarray'] = nd
    pkg = importlib.import_module(__name__.rsplit('.', 1)[0])
    op = importlib.import_module(pkg.__name__ + '.operator_py')
    sys.modules.setdefault('operator_py', op)
    for sub, real in (('proposal', 'proposal'), ('proposal_target', 'targets'), ('box_annotator_ohem', 'targets'),
                      ('nms_multi_target', 'targets'), ('learn_nms', 'learn_nms')):
        sys.modules.setdefault('operator_py.' + sub, importlib.import_module('%s.operator_py.%s' % (pkg.__name__, real)))
    utils = sys.modules.setdefault('utils', types.ModuleType('utils'))
    if not hasattr(utils, '__path__'):
        utils.__path__ = []
    sys.modules.setdefault('utils.symbol', importlib.import_module(pkg.__name__ + '.utils_symbol'))
    utils.symbol = sys.modules['utils.symbol']
    for name in ('nms', 'bbox'):
        sys.modules.setdefault(name, importlib.import_module('%s.%s' % (pkg.__name__, name)))
    sys.modules.setdefault('nms.nms', importlib.import_module(pkg.__name__ + '.nms.nms'))
    sys.modules.setdefault('bbox.bbox', importlib.import_module(pkg.__name__ + '.bbox.bbox'))
    if py2_shims:
        import builtins
        import pickle
        import numpy as np
        sys.modules.setdefault('cPickle', pickle)
        if not hasattr(builtins, 'xrange'):
            builtins.xrange = range
        for alias, tp in (('float', float), ('int', int), ('bool', bool)):
            if not hasattr(np, alias):
                setattr(np, alias, tp)
    return me


_bind_operator_namespace()
