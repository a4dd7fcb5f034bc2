"""Symbolic graph builder with the `mx.sym` surface the reference's graph files use
(relation_rcnn/symbols/*.py; base class lib/utils/symbol.py:10-56): `Variable`, operator
constructors with MXNet's composition rules, `Group`, arithmetic overloads, `list_arguments`,
`list_outputs`, `list_auxiliary_states`, `infer_shape`, `get_internals`, `tojson`.

What is mirrored from MXNet v1.1.0 (library not vendored in the reference: semantics restated, unpinned):
  * every operator has an ordered list of tensor inputs; inputs not supplied are created as variables
    `<name>_<input>` (`fc_new_1_weight`, `bn2a_branch1_moving_mean`, `rpn_cls_prob_label`);
  * unnamed operators get `<opname.lower()><counter>` names;
  * outputs are `<name>_output` (`<name>_output<i>` / `<name>_<outname>` for multi-output operators);
  * `list_arguments()` = variables in depth-first post-order of the inputs; BatchNorm's moving statistics are
    auxiliary states;
  * every attribute is kept as given and stringified only for `Custom` operators (MXNet hands CustomOpProp
    constructors strings, operator_py/proposal.py:202-212).
Python-3 note: the reference is Python-2 code; `dim[0] / group` (SYM_REL:104) is an int there and a float here,
so integral floats inside shape-like attributes are converted back to ints.
"""
import json

from . import registry as R


class _NameManager(object):
    def __init__(self):
        self.counter = {}

    def get(self, name, hint):
        if name:
            return name
        n = self.counter.get(hint, 0)
        self.counter[hint] = n + 1
        return '%s%d' % (hint, n)


_names = _NameManager()


def reset_names():
    """Restart the automatic operator numbering (a fresh graph in a fresh process would start at 0)."""
    _names.counter.clear()


class Node(object):
    __slots__ = ('op', 'name', 'attrs', 'inputs', 'num_outputs', 'out_names', 'is_aux')

    def __init__(self, op, name, attrs, inputs, num_outputs=1, out_names=None):
        self.op, self.name, self.attrs, self.inputs = op, name, attrs, inputs
        self.num_outputs = num_outputs
        self.out_names = out_names
        self.is_aux = False

    def output_name(self, i):
        if self.op == 'null':
            return self.name
        if self.out_names is not None:
            return '%s_%s' % (self.name, self.out_names[i])
        return '%s_output' % self.name if self.num_outputs == 1 else '%s_output%d' % (self.name, i)


class Symbol(object):
    """A list of (node, output index) heads -- one entry for an operator output, several for a Group or a
    multi-output operator."""

    def __init__(self, heads):
        self.heads = list(heads)

    # ---- composition ------------------------------------------------------------------------------------
    @property
    def name(self):
        return self.heads[0][0].name if len(self.heads) == 1 else None

    def __len__(self):
        return len(self.heads)

    def __iter__(self):
        return (Symbol([h]) for h in self.heads)

    def __getitem__(self, k):
        if isinstance(k, str):
            names = self.list_outputs()
            if k not in names:
                raise ValueError("no output named %r (have %d outputs)" % (k, len(names)))
            k = names.index(k)
        return Symbol([self.heads[k]])

    def _scalar_or_sym(self, other, op, scalar_op, rev_scalar_op=None, rev=False):
        if isinstance(other, Symbol):
            a, b = (other, self) if rev else (self, other)
            return R.make(op, [a, b], {})
        sop = rev_scalar_op if (rev and rev_scalar_op) else scalar_op
        return R.make(sop, [self], {'scalar': float(other)})

    def __add__(self, o): return self._scalar_or_sym(o, '_plus', '_plus_scalar')
    def __radd__(self, o): return self._scalar_or_sym(o, '_plus', '_plus_scalar', rev=True)
    def __sub__(self, o): return self._scalar_or_sym(o, '_minus', '_minus_scalar')
    def __rsub__(self, o): return self._scalar_or_sym(o, '_minus', '_minus_scalar', '_rminus_scalar', rev=True)
    def __mul__(self, o): return self._scalar_or_sym(o, '_mul', '_mul_scalar')
    def __rmul__(self, o): return self._scalar_or_sym(o, '_mul', '_mul_scalar', rev=True)
    def __truediv__(self, o): return self._scalar_or_sym(o, '_div', '_div_scalar')
    def __rtruediv__(self, o): return self._scalar_or_sym(o, '_div', '_div_scalar', '_rdiv_scalar', rev=True)
    __div__, __rdiv__ = __truediv__, __rtruediv__
    def __neg__(self): return self * -1.0
    def __pow__(self, o): return self._scalar_or_sym(o, '_power', '_power_scalar')
    def __rpow__(self, o): return self._scalar_or_sym(o, '_power', '_power_scalar', '_rpower_scalar', rev=True)

    # ---- graph queries ----------------------------------------------------------------------------------
    def _topo(self):
        """Nodes in depth-first post-order over the inputs (MXNet's DFSVisit order)."""
        seen, order = set(), []
        stack = [(n, False) for n, _ in reversed(self.heads)]
        while stack:
            node, done = stack.pop()
            if done:
                order.append(node)
                continue
            if id(node) in seen:
                continue
            seen.add(id(node))
            stack.append((node, True))
            for src, _ in reversed(node.inputs):
                if id(src) not in seen:
                    stack.append((src, False))
        return order

    def list_arguments(self):
        return [n.name for n in self._topo() if n.op == 'null' and not n.is_aux]

    def list_auxiliary_states(self):
        return [n.name for n in self._topo() if n.op == 'null' and n.is_aux]

    def list_outputs(self):
        return [n.output_name(i) for n, i in self.heads]

    def list_inputs(self):
        return [n.name for n in self._topo() if n.op == 'null']

    def get_internals(self):
        heads = []
        for n in self._topo():
            heads.extend((n, i) for i in range(n.num_outputs if n.op != 'null' else 1))
        return Symbol(heads)

    def get_children(self):
        if len(self.heads) != 1:
            return None
        return Symbol(list(self.heads[0][0].inputs)) if self.heads[0][0].inputs else None

    def attr(self, key):
        v = self.heads[0][0].attrs.get(key)
        return None if v is None else str(v)

    def infer_shape(self, *args, **kwargs):
        """(arg_shapes, out_shapes, aux_shapes) in list_arguments / list_outputs / list_auxiliary_states order."""
        from . import executor
        if args:
            kwargs = dict(zip(self.list_arguments(), args))
        shapes = executor.infer_shapes(self, {k: tuple(v) for k, v in kwargs.items() if v is not None})
        return ([shapes['var'].get(n) for n in self.list_arguments()],
                [shapes['out'][(id(n), i)] for n, i in self.heads],
                [shapes['var'].get(n) for n in self.list_auxiliary_states()])

    def infer_shape_partial(self, *args, **kwargs):
        return self.infer_shape(*args, **kwargs)

    def bind(self, ctx=None, args=None, args_grad=None, grad_req='null', aux_states=None, **kw):
        from . import executor
        return executor.Executor(self, args or {}, aux_states or {}, ctx=ctx, **kw)

    def simple_bind(self, ctx=None, grad_req='null', **shapes):
        raise NotImplementedError("bind() with explicit arrays is the supported form")

    def tojson(self):
        """Graph as JSON in MXNet's node-list layout (nodes / arg_nodes / heads); attributes are stringified."""
        order = self._topo()
        index = {id(n): i for i, n in enumerate(order)}
        nodes = []
        for n in order:
            d = {'op': n.op, 'name': n.name, 'inputs': [[index[id(s)], i, 0] for s, i in n.inputs]}
            if n.attrs:
                d['attrs'] = {k: str(v) for k, v in n.attrs.items()}
            if n.is_aux:
                d['is_aux'] = True
            if n.op != 'null' and (n.num_outputs != 1 or n.out_names):
                d['num_outputs'] = n.num_outputs
                if n.out_names:
                    d['out_names'] = list(n.out_names)
            nodes.append(d)
        return json.dumps({'nodes': nodes, 'arg_nodes': [i for i, n in enumerate(order) if n.op == 'null'],
                           'heads': [[index[id(n)], i, 0] for n, i in self.heads],
                           'attrs': {'relnet_amd_mx': 1}}, indent=1)

    def save(self, fname):
        with open(fname, 'w') as f:
            f.write(self.tojson())

    def __repr__(self):
        return '<Symbol %s>' % (self.name if len(self.heads) == 1 else 'group[%d]' % len(self.heads))


def load_json(text):
    """Inverse of Symbol.tojson(): attributes come back as strings and are parsed on use, exactly as
    operators receive them from a `-symbol.json` checkpoint."""
    g = json.loads(text)
    nodes = []
    for d in g['nodes']:
        n = Node(d['op'], d['name'], dict(d.get('attrs', {})), [(nodes[s], i) for s, i, _ in d['inputs']],
                 d.get('num_outputs', 1), d.get('out_names'))
        n.is_aux = bool(d.get('is_aux', False))
        nodes.append(n)
    return Symbol([(nodes[i], j) for i, j, _ in g['heads']])


def load(fname):
    with open(fname) as f:
        return load_json(f.read())


def Variable(name, attr=None, shape=None, lr_mult=None, wd_mult=None, dtype=None, init=None, **kwargs):
    attrs = {}
    if shape is not None:
        attrs['__shape__'] = tuple(shape)
    if lr_mult is not None:
        attrs['__lr_mult__'] = lr_mult
    if wd_mult is not None:
        attrs['__wd_mult__'] = wd_mult
    return Symbol([(Node('null', name, attrs, []), 0)])


var = Variable


def Group(symbols):
    heads = []
    for s in symbols:
        heads.extend(s.heads)
    return Symbol(heads)
