"""Operator table of the `mx` facade: for every MXNet operator the reference's graph files use
(SURVEY.md 8(b) row 4), its ordered tensor inputs, number of outputs, the shapes of the parameters it creates,
and its evaluation on torch tensors.

Two kinds of evaluation functions:
  * LIGHT operators (reshapes, slices, broadcasts, elementwise math, small reductions, sort / take) are torch
    tensor expressions -- device plumbing that also runs on `meta` tensors, which is how `infer_shape` works;
  * HEAVY operators (Convolution, FullyConnected, dot / batch_dot, ROIPooling, Pooling, Custom, the deformable
    contrib operators) have a `meta` shape rule here and NO torch arithmetic: the executor dispatches them to
    librelnet_hip.so (mx/executor.py) and refuses to run them anywhere else.
Operator semantics restated from the MXNet v1.1.0 operator documentation (unpinned; the same restatement as the
numpy stand-in the goldens were generated with, tests/golden/refshim).
"""
import ast
import math

import torch

from .graph import Symbol, Node, _names

OPS = {}


# ---- attribute parsing (values arrive as python objects from graph code, as strings from JSON) -------------
def _lit(v):
    if isinstance(v, str):
        s = v.strip()
        if s in ('None', ''):
            return None
        if s in ('True', 'true'):
            return True
        if s in ('False', 'false'):
            return False
        try:
            return ast.literal_eval(s)
        except (ValueError, SyntaxError):
            return s
    return v


def a_get(attrs, key, default=None):
    return _lit(attrs[key]) if key in attrs and attrs[key] is not None else default


def a_int(attrs, key, default=None):
    v = a_get(attrs, key, default)
    return None if v is None else int(v)


def a_float(attrs, key, default=None):
    v = a_get(attrs, key, default)
    return None if v is None else float(v)


def a_bool(attrs, key, default=False):
    v = a_get(attrs, key, default)
    return bool(v)


def a_tuple(attrs, key, default=None):
    v = a_get(attrs, key, default)
    if v is None:
        return None
    if isinstance(v, (int, float)):
        v = (v,)
    return tuple(None if x is None else int(x) for x in v)          # integral floats (py2 `/`) -> ints


def a_str(attrs, key, default=None):
    v = attrs.get(key, default)
    return default if v is None else str(v)


class OpDef(object):
    def __init__(self, name, inputs, fn=None, nout=1, params=None, aux=(), out_names=None, variadic=False, heavy=False,
                 hint=None):
        self.name, self.inputs, self.fn, self.nout = name, inputs, fn, nout
        self.params = params            # callable(attrs, data_shapes: dict) -> {input_name: shape}
        self.aux = set(aux)
        self.out_names, self.variadic, self.heavy = out_names, variadic, heavy
        self.hint = hint or name.lower().lstrip('_')

    def inputs_for(self, attrs):
        ins = self.inputs(attrs) if callable(self.inputs) else list(self.inputs)
        return ins

    def num_outputs(self, attrs):
        return self.nout(attrs) if callable(self.nout) else self.nout


def defop(name, inputs, fn=None, **kw):
    OPS[name] = OpDef(name, inputs, fn, **kw)
    return OPS[name]


def make(op, args, kwargs):
    """Compose an operator node from positional / keyword symbols and attributes (MXNet's `_compose`)."""
    d = OPS[op]
    kwargs = dict(kwargs)
    name = kwargs.pop('name', None)
    kwargs.pop('attr', None)
    sym_kw = {k: v for k, v in kwargs.items() if isinstance(v, Symbol)}
    attrs = {k: v for k, v in kwargs.items() if not isinstance(v, Symbol) and v is not None or k in ('end',)}
    attrs = {k: v for k, v in attrs.items() if not isinstance(v, Symbol)}
    pos_attrs = getattr(d, 'pos_attrs', None)
    if pos_attrs and args and not isinstance(args[0], Symbol):          # mx.sym.full((1,), 1000.0), mx.sym.arange(0, 8)
        for k_, v_ in zip(pos_attrs, args):
            attrs[k_] = v_
        args = ()
    if d.variadic:
        ins = list(args) + [sym_kw[k] for k in sorted(sym_kw)]
        attrs.setdefault('num_args', len(ins))
        name = _names.get(name, d.hint)
        heads = []
        for s in ins:
            if len(s.heads) != 1:
                raise ValueError("%s: every input must be a single-output symbol" % op)
            heads.append(s.heads[0])
        node = Node(op, name, attrs, heads, d.num_outputs(attrs), d.out_names)
        return Symbol([(node, i) for i in range(node.num_outputs)])
    in_names = d.inputs_for(attrs)
    name = _names.get(name, d.hint)
    given = {}
    for n, s in zip(in_names, args):
        given[n] = s
    for k, s in sym_kw.items():
        if k not in in_names:
            raise TypeError("%s got an unexpected tensor argument %r (inputs: %s)" % (op, k, in_names))
        given[k] = s
    heads = []
    for n in in_names:
        if n in given:
            s = given[n]
            if len(s.heads) != 1:
                raise ValueError("%s: input %r must be a single-output symbol" % (op, n))
            heads.append(s.heads[0])
        else:                               # auto-created parameter / label / auxiliary state
            v = Node('null', '%s_%s' % (name, n), {}, [])
            v.is_aux = n in d.aux
            heads.append((v, 0))
    node = Node(op, name, attrs, heads, d.num_outputs(attrs), d.out_names)
    return Symbol([(node, i) for i in range(node.num_outputs)])


# ---------------------------------------------------------------------------------------------------------
# shapes
# ---------------------------------------------------------------------------------------------------------
def reshape_codes(in_shape, codes, reverse=False):
    """MXNet Reshape special codes 0 (copy), -1 (infer), -2 (copy the rest), -3 (merge two), -4 (split)."""
    if reverse:       # match the codes from the right: reverse shape and codes, keeping every -4 with its two arguments
        codes = [int(c) for c in codes]
        units, k = [], 0
        while k < len(codes):
            if codes[k] == -4:
                units.append([-4, codes[k + 2], codes[k + 1]]); k += 3        # the split's factors swap with the axis order
            else:
                units.append([codes[k]]); k += 1
        rev = [c for u in reversed(units) for c in u]
        return tuple(reversed(reshape_codes(tuple(reversed(in_shape)), tuple(rev))))
    out, i, infer, k = [], 0, None, 0
    codes = [int(c) for c in codes]
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
        elif c == -4:
            d1, d2 = codes[k + 1], codes[k + 2]
            if d1 == -1:
                d1 = in_shape[i] // d2
            if d2 == -1:
                d2 = in_shape[i] // d1
            out.extend([d1, d2]); i += 1; k += 2
        else:
            raise ValueError("bad Reshape code %d" % c)
        k += 1
    if infer is not None:
        known = 1
        for j, v in enumerate(out):
            if j != infer:
                known *= v
        total = 1
        for v in in_shape:
            total *= v
        out[infer] = total // known if known else 0
    return tuple(int(v) for v in out)


def conv_out(n, k, s, p, d):
    return (n + 2 * p - d * (k - 1) - 1) // s + 1


def _meta(shape, like=None, dtype=None):
    return torch.empty(tuple(int(s) for s in shape), device='meta', dtype=dtype or (like.dtype if like is not None else torch.float32))


# ---------------------------------------------------------------------------------------------------------
# light operators
# ---------------------------------------------------------------------------------------------------------
def _ew(name, f):
    defop(name, ['data'], lambda a, x: f(x))


for _n, _f in (('abs', torch.abs), ('exp', torch.exp), ('sqrt', torch.sqrt), ('zeros_like', torch.zeros_like),
               ('ones_like', torch.ones_like), ('BlockGrad', lambda x: x), ('identity', lambda x: x),
               ('negative', torch.neg), ('sigmoid', torch.sigmoid), ('relu', torch.relu)):
    _ew(_n, _f)
OPS['BlockGrad'].hint = 'blockgrad'


def _precise(f64):
    """log / sin / cos as correctly rounded float32 (evaluated in float64, rounded once): the relation logits are
    ill-conditioned in these, and MXNet's own fp32 kernels are faithful to < 1 ulp."""
    def g(a, x):
        if x.dtype == torch.float32 and x.device.type != 'meta':
            return f64(x.double()).float()
        return f64(x)
    return g


defop('log', ['data'], _precise(torch.log))
defop('sin', ['data'], _precise(torch.sin))
defop('cos', ['data'], _precise(torch.cos))

for _n, _f in (('_plus', torch.add), ('_minus', torch.sub), ('_mul', torch.mul), ('_div', torch.div),
               ('_power', torch.pow), ('_maximum', torch.maximum), ('_minimum', torch.minimum),
               ('broadcast_add', torch.add), ('broadcast_plus', torch.add), ('broadcast_minus', torch.sub),
               ('broadcast_sub', torch.sub), ('broadcast_mul', torch.mul), ('broadcast_div', torch.div),
               ('broadcast_power', torch.pow), ('broadcast_maximum', torch.maximum),
               ('broadcast_minimum', torch.minimum), ('elemwise_add', torch.add), ('elemwise_mul', torch.mul)):
    defop(_n, ['lhs', 'rhs'], (lambda f: lambda a, x, y: f(x, y))(_f))

for _n, _f in (('_plus_scalar', lambda x, s: x + s), ('_minus_scalar', lambda x, s: x - s),
               ('_rminus_scalar', lambda x, s: s - x), ('_mul_scalar', lambda x, s: x * s),
               ('_div_scalar', lambda x, s: x / s), ('_rdiv_scalar', lambda x, s: s / x),
               ('_power_scalar', lambda x, s: torch.pow(x, s)),
               ('_rpower_scalar', lambda x, s: torch.pow(torch.as_tensor(s, dtype=x.dtype, device=x.device), x)),
               ('_maximum_scalar', lambda x, s: torch.clamp(x, min=s)),
               ('_minimum_scalar', lambda x, s: torch.clamp(x, max=s))):
    defop(_n, ['data'], (lambda f: lambda a, x: f(x, a_float(a, 'scalar')))(_f))


def _activation(a, x):
    t = a_str(a, 'act_type')
    if t == 'relu':
        return torch.relu(x)
    if t == 'sigmoid':
        return torch.sigmoid(x)
    if t == 'tanh':
        return torch.tanh(x)
    raise NotImplementedError("Activation act_type=%r" % t)


defop('Activation', ['data'], _activation)
defop('Reshape', ['data'], lambda a, x: x.reshape(reshape_codes(tuple(x.shape), a_tuple(a, 'shape'), a_bool(a, 'reverse'))))
defop('Flatten', ['data'], lambda a, x: x.reshape(x.shape[0], -1))
defop('transpose', ['data'], lambda a, x: x.permute(*(a_tuple(a, 'axes') or tuple(reversed(range(x.dim()))))))
defop('expand_dims', ['data'], lambda a, x: x.unsqueeze(a_int(a, 'axis')))


def _slice_axis(a, x):
    ax, b, e = a_int(a, 'axis'), a_int(a, 'begin', 0), a_get(a, 'end')
    n = x.shape[ax]
    e = n if e is None else int(e)
    b, e = (b + n if b < 0 else b), (e + n if e < 0 else e)
    return x.narrow(ax, b, e - b)


defop('slice_axis', ['data'], _slice_axis)


def _slice(a, x):
    begin, end = a_get(a, 'begin'), a_get(a, 'end')
    idx = []
    for d, (b, e) in enumerate(zip(begin, end)):
        idx.append(slice(None if b is None else int(b), None if e is None else int(e)))
    return x[tuple(idx)]


defop('slice', ['data'], _slice)


def _split(a, x):
    n, ax = a_int(a, 'num_outputs'), a_int(a, 'axis', 1)
    parts = torch.chunk(x, n, dim=ax)
    return [p.squeeze(ax) for p in parts] if a_bool(a, 'squeeze_axis') else list(parts)


defop('split', ['data'], _split, nout=lambda a: a_int(a, 'num_outputs'))
defop('SliceChannel', ['data'], _split, nout=lambda a: a_int(a, 'num_outputs'), hint='slicechannel')
defop('Concat', [], lambda a, *xs: torch.cat(xs, dim=a_int(a, 'dim', 1)), variadic=True, hint='concat')
defop('concat', [], lambda a, *xs: torch.cat(xs, dim=a_int(a, 'dim', 1)), variadic=True)
defop('ElementWiseSum', [], lambda a, *xs: sum(xs[1:], xs[0]), variadic=True, hint='elementwisesum')
defop('add_n', [], lambda a, *xs: sum(xs[1:], xs[0]), variadic=True)


def _broadcast_to(a, x):
    shape = a_tuple(a, 'shape')
    return x.expand(*[x.shape[i] if s == 0 else s for i, s in enumerate(shape)])


defop('broadcast_to', ['data'], _broadcast_to)
defop('tile', ['data'], lambda a, x: (lambda r: x.repeat(*((1,) * (x.dim() - len(r)) + tuple(r))))(a_tuple(a, 'reps')))     # MXNet left-pads reps with 1
defop('reverse', ['data'], lambda a, x: torch.flip(x, dims=list(a_tuple(a, 'axis'))))
defop('flip', ['data'], lambda a, x: torch.flip(x, dims=list(a_tuple(a, 'axis'))))
defop('where', ['condition', 'x', 'y'], lambda a, c, x, y: torch.where(c != 0, x, y))


def _red(f):
    def g(a, x):
        ax = a_tuple(a, 'axis')
        kd = a_bool(a, 'keepdims')
        if ax is None:
            return f(x, tuple(range(x.dim())), kd)
        return f(x, ax, kd)
    return g


defop('mean', ['data'], _red(lambda x, ax, kd: x.mean(dim=ax, keepdim=kd)))
defop('sum', ['data'], _red(lambda x, ax, kd: x.sum(dim=ax, keepdim=kd)))
defop('max', ['data'], _red(lambda x, ax, kd: x.amax(dim=ax, keepdim=kd)))
defop('min', ['data'], _red(lambda x, ax, kd: x.amin(dim=ax, keepdim=kd)))


def _device_of(ctx):
    return ctx.get('device', 'meta') if isinstance(ctx, dict) else 'meta'


def _arange(a, ctx=None):
    start, stop, step, rep = a_float(a, 'start', 0.0), a_get(a, 'stop'), a_float(a, 'step', 1.0), a_int(a, 'repeat', 1)
    if stop is None:
        start, stop = 0.0, start
    n = max(int(math.ceil((float(stop) - start) / step)), 0)
    # MXNet's range kernel: out[i] = start + (i / repeat) * step in float32
    v = torch.arange(n, device=_device_of(ctx), dtype=torch.float32) * torch.tensor(step, dtype=torch.float32).item() + start
    return v.repeat_interleave(rep) if rep > 1 else v


defop('arange', [], _arange, hint='arange')
OPS['arange'].no_input = True
defop('full', [], lambda a, ctx=None: torch.full(a_tuple(a, 'shape'), a_float(a, 'val'), device=_device_of(ctx), dtype=torch.float32))
defop('zeros', [], lambda a, ctx=None: torch.zeros(a_tuple(a, 'shape'), device=_device_of(ctx), dtype=torch.float32))
defop('ones', [], lambda a, ctx=None: torch.ones(a_tuple(a, 'shape'), device=_device_of(ctx), dtype=torch.float32))
for _n in ('full', 'zeros', 'ones'):
    OPS[_n].no_input = True
OPS['full'].pos_attrs = ('shape', 'val')
OPS['zeros'].pos_attrs = OPS['ones'].pos_attrs = ('shape',)
OPS['arange'].pos_attrs = ('start', 'stop', 'step', 'repeat')


def _take(a, x, idx):
    ax = a_int(a, 'axis', 0)
    i = idx.long().clamp(0, x.shape[ax] - 1)                       # mode='clip'
    out = torch.index_select(x, ax, i.reshape(-1))
    return out.reshape(tuple(x.shape[:ax]) + tuple(idx.shape) + tuple(x.shape[ax + 1:]))


defop('take', ['a', 'indices'], _take)


def _pick(a, x, idx):
    ax = a_int(a, 'axis', -1)
    ax = ax + x.dim() if ax < 0 else ax
    i = idx.long().clamp(0, x.shape[ax] - 1).unsqueeze(ax)
    out = torch.gather(x, ax, i)
    return out if a_bool(a, 'keepdims') else out.squeeze(ax)


defop('pick', ['data', 'index'], _pick)


def _sort(a, x):
    return torch.sort(x, dim=a_int(a, 'axis', -1), descending=not a_bool(a, 'is_ascend', True), stable=True)[0]


def _argsort(a, x):
    # equal keys: MXNet's device sort is not stable; this facade orders ties by ascending index (the HIP learn-NMS
    # path documents its own rule; inputs of the parity tests are tie-free)
    return torch.sort(x, dim=a_int(a, 'axis', -1), descending=not a_bool(a, 'is_ascend', True), stable=True)[1].to(torch.float32)


defop('sort', ['data'], _sort)
defop('argsort', ['data'], _argsort)
defop('softmax', ['data'], lambda a, x: torch.softmax(x, dim=a_int(a, 'axis', -1)))


def _softmax_activation(a, x):
    if a_str(a, 'mode', 'instance') == 'channel':
        return torch.softmax(x, dim=1)
    return torch.softmax(x.reshape(x.shape[0], -1), dim=1).reshape(x.shape)


defop('SoftmaxActivation', ['data'], _softmax_activation, hint='softmaxactivation')


def _softmax_output(a, x, label):
    if a_bool(a, 'multi_output'):
        return torch.softmax(x, dim=1)
    return torch.softmax(x.reshape(x.shape[0], -1), dim=1).reshape(x.shape)


def _softmax_output_params(a, shapes):
    d = shapes['data']
    if a_bool(a, 'multi_output'):
        return {'label': (d[0],) + tuple(d[2:])}
    return {'label': (d[0],)}


defop('SoftmaxOutput', ['data', 'label'], _softmax_output, params=_softmax_output_params, hint='softmaxoutput')
defop('Softmax', ['data', 'label'], _softmax_output, params=_softmax_output_params)


def _smooth_l1(a, x):
    s2 = a_float(a, 'scalar', 1.0) ** 2
    ax = x.abs()
    return torch.where(ax < 1.0 / s2, 0.5 * s2 * x * x, ax - 0.5 / s2)


defop('smooth_l1', ['data'], _smooth_l1)
defop('MakeLoss', ['data'], lambda a, x: x, hint='makeloss')
defop('make_loss', ['data'], lambda a, x: x)


def _upsampling(a, *xs):
    if a_str(a, 'sample_type', 'nearest') != 'nearest' or len(xs) != 1:
        raise NotImplementedError("UpSampling: nearest with one input only")
    s = a_int(a, 'scale')
    return xs[0].repeat_interleave(s, dim=2).repeat_interleave(s, dim=3)


defop('UpSampling', [], _upsampling, variadic=True, hint='upsampling')


def _crop(a, *xs):
    x = xs[0]
    off = a_tuple(a, 'offset', (0, 0))
    if len(xs) == 2:
        h, w = xs[1].shape[2], xs[1].shape[3]
    else:
        h, w = a_tuple(a, 'h_w')
    if a_bool(a, 'center_crop'):
        off = ((x.shape[2] - h) // 2, (x.shape[3] - w) // 2)
    return x[:, :, off[0]:off[0] + h, off[1]:off[1] + w]


defop('Crop', [], _crop, variadic=True, hint='crop')


# ---------------------------------------------------------------------------------------------------------
# heavy operators: shape rules only (the executor runs them on the HIP library)
# ---------------------------------------------------------------------------------------------------------
def _conv_inputs(a):
    return ['data', 'weight'] + ([] if a_bool(a, 'no_bias') else ['bias'])


def _conv_params(a, shapes):
    c = shapes['data'][1]
    k = a_tuple(a, 'kernel')
    nf, g = a_int(a, 'num_filter'), a_int(a, 'num_group', 1)
    return {'weight': (nf, c // g) + tuple(k), 'bias': (nf,)}


def _conv_meta(a, x, w, b=None):
    k, s, d, p = a_tuple(a, 'kernel'), a_tuple(a, 'stride', (1, 1)), a_tuple(a, 'dilate', (1, 1)), a_tuple(a, 'pad', (0, 0))
    return _meta((x.shape[0], a_int(a, 'num_filter'), conv_out(x.shape[2], k[0], s[0], p[0], d[0]),
                  conv_out(x.shape[3], k[1], s[1], p[1], d[1])), x)


defop('Convolution', _conv_inputs, _conv_meta, params=_conv_params, heavy=True, hint='convolution')


def _bn_params(a, shapes):
    c = shapes['data'][1]
    return {'gamma': (c,), 'beta': (c,), 'moving_mean': (c,), 'moving_var': (c,)}


def _batchnorm(a, x, gamma, beta, mean, var):
    """Inference form only (use_global_stats=True everywhere in the reference, SYM_BASE:32): a frozen affine."""
    if not a_bool(a, 'use_global_stats'):
        raise NotImplementedError("BatchNorm with batch statistics is not on this path (use_global_stats=True)")
    eps = a_float(a, 'eps', 1e-3)
    g = torch.ones_like(gamma) if a_bool(a, 'fix_gamma', True) else gamma
    shape = (1, -1) + (1,) * (x.dim() - 2)
    s = (g.double() / torch.sqrt(var.double() + eps)).to(x.dtype) if x.device.type != 'meta' else g
    t = (beta.double() - mean.double() * (g.double() / torch.sqrt(var.double() + eps))).to(x.dtype) if x.device.type != 'meta' else beta
    return x * s.reshape(shape) + t.reshape(shape)


defop('BatchNorm', ['data', 'gamma', 'beta', 'moving_mean', 'moving_var'], _batchnorm, params=_bn_params,
      aux=('moving_mean', 'moving_var'), hint='batchnorm')


def _fc_inputs(a):
    return ['data', 'weight'] + ([] if a_bool(a, 'no_bias') else ['bias'])


def _fc_params(a, shapes):
    d = shapes['data']
    k = 1
    for v in d[1:]:
        k *= v
    if not a_bool(a, 'flatten', True):
        k = d[-1]
    return {'weight': (a_int(a, 'num_hidden'), k), 'bias': (a_int(a, 'num_hidden'),)}


def _fc_meta(a, x, w, b=None):
    if a_bool(a, 'flatten', True):
        return _meta((x.shape[0], a_int(a, 'num_hidden')), x)
    return _meta(tuple(x.shape[:-1]) + (a_int(a, 'num_hidden'),), x)


defop('FullyConnected', _fc_inputs, _fc_meta, params=_fc_params, heavy=True, hint='fullyconnected')


def _pool_meta(a, x):
    if a_bool(a, 'global_pool'):
        return _meta((x.shape[0], x.shape[1], 1, 1), x)
    k, s, p = a_tuple(a, 'kernel'), a_tuple(a, 'stride', (1, 1)), a_tuple(a, 'pad', (0, 0))
    full = a_str(a, 'pooling_convention', 'valid') == 'full'

    def o(n, k_, s_, p_):
        v = (n + 2 * p_ - k_)
        return (-(-v // s_) if full else v // s_) + 1
    return _meta((x.shape[0], x.shape[1], o(x.shape[2], k[0], s[0], p[0]), o(x.shape[3], k[1], s[1], p[1])), x)


defop('Pooling', ['data'], _pool_meta, heavy=True, hint='pooling')


def _dot_meta(a, x, y):
    ta, tb = a_bool(a, 'transpose_a'), a_bool(a, 'transpose_b')
    m = x.shape[1] if ta else x.shape[0]
    n = y.shape[0] if tb else y.shape[-1]
    return _meta((m,) + ((n,) if y.dim() > 1 else ()), x)


def _batch_dot_meta(a, x, y):
    ta, tb = a_bool(a, 'transpose_a'), a_bool(a, 'transpose_b')
    return _meta((x.shape[0], x.shape[2] if ta else x.shape[1], y.shape[1] if tb else y.shape[2]), x)


defop('dot', ['lhs', 'rhs'], _dot_meta, heavy=True)
defop('batch_dot', ['lhs', 'rhs'], _batch_dot_meta, heavy=True)


def _roipool_meta(a, x, rois):
    ps = a_tuple(a, 'pooled_size')
    return _meta((rois.shape[0], x.shape[1], ps[0], ps[1]), x)


defop('ROIPooling', ['data', 'rois'], _roipool_meta, heavy=True, hint='roipooling')
# mx.contrib.sym.ROIAlign(data, rois, pooled_size, spatial_scale, sample_ratio) (MXNet >= 1.3; not used by the reference's symbols)
defop('_contrib_ROIAlign', ['data', 'rois'], _roipool_meta, heavy=True, hint='roialign')


def _dconv_inputs(a):
    return ['data', 'offset', 'weight'] + ([] if a_bool(a, 'no_bias') else ['bias'])


defop('_contrib_DeformableConvolution', _dconv_inputs, lambda a, x, off, w, b=None: _conv_meta(a, x, w, b),
      params=_conv_params, heavy=True, hint='deformableconvolution')


def _dpsroi_inputs(a):
    return ['data', 'rois'] + ([] if a_bool(a, 'no_trans') else ['trans'])


def _dpsroi_meta(a, x, rois, trans=None):
    p = a_int(a, 'pooled_size')
    return _meta((rois.shape[0], a_int(a, 'output_dim'), p, p), x)


defop('_contrib_DeformablePSROIPooling', _dpsroi_inputs, _dpsroi_meta, heavy=True, hint='deformablepsroipooling')


def _proposal_meta(a, cls_prob, bbox_pred, im_info):
    n = a_int(a, 'rpn_post_nms_top_n', 300)
    outs = [_meta((n, 5), cls_prob)]
    if a_bool(a, 'output_score'):
        outs.append(_meta((n, 1), cls_prob))
    return outs if len(outs) > 1 else outs[0]


defop('_contrib_Proposal', ['cls_prob', 'bbox_pred', 'im_info'], _proposal_meta, heavy=True,
      nout=lambda a: 2 if a_bool(a, 'output_score') else 1, hint='proposal')


# ---- Custom: inputs / outputs come from the registered CustomOpProp --------------------------------------
def custom_prop(attrs):
    from .. import operator_py
    kw = {k: str(v) for k, v in attrs.items() if k not in ('op_type', 'num_args')}
    return operator_py.get_prop(str(attrs['op_type']))(**kw)


def make_custom(args, kwargs):
    kwargs = dict(kwargs)
    name = kwargs.pop('name', None)
    sym_kw = {k: v for k, v in kwargs.items() if isinstance(v, Symbol)}
    attrs = {k: v for k, v in kwargs.items() if not isinstance(v, Symbol)}
    prop = custom_prop(attrs)
    in_names, out_names = prop.list_arguments(), prop.list_outputs()
    name = _names.get(name, 'custom')
    given = dict(zip(in_names, args))
    given.update(sym_kw)
    heads = []
    for n in in_names:
        if n in given:
            heads.append(given[n].heads[0])
        else:
            heads.append((Node('null', '%s_%s' % (name, n), {}, []), 0))
    node = Node('Custom', name, attrs, heads, len(out_names), list(out_names))
    return Symbol([(node, i) for i in range(node.num_outputs)])


def _custom_meta(a, *xs):
    prop = custom_prop(a)
    res = prop.infer_shape([tuple(x.shape) for x in xs])
    outs = [_meta(s, xs[0]) for s in res[1]]
    return outs if len(outs) > 1 else outs[0]


def _custom_params(a, shapes_list):
    """Custom ops complete their own input shapes (e.g. learn_nms's weight arguments)."""
    prop = custom_prop(a)
    return prop.infer_shape(shapes_list)[0]


OPS['Custom'] = OpDef('Custom', lambda a: custom_prop(a).list_arguments(), _custom_meta, heavy=True,
                      nout=lambda a: len(custom_prop(a).list_outputs()), hint='custom')
OPS['Custom'].custom_params = _custom_params
