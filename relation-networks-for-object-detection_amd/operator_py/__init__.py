"""Mirror of the reference's Python CustomOp protocol (mx.operator.CustomOp /
CustomOpProp / register, as used by relation_rcnn/operator_py/*.py) over torch device
tensors.  Operators keep the reference's registered names, argument/ output lists, string
attribute parsing and error behaviour; their forward() calls the HIP kernels.
"""
import torch

_REGISTRY = {}


class CustomOp(object):
    """forward(is_train, req, in_data, out_data, aux) / backward(...); results are written
    with self.assign(dst, req, src) where req is 'null' | 'write' | 'inplace' | 'add'."""

    def assign(self, dst, req, src):
        if req == 'null':
            return
        if not torch.is_tensor(src):
            src = torch.full_like(dst, float(src))
        if req == 'add':
            dst.add_(src.to(dst.dtype).reshape(dst.shape))
        elif req in ('write', 'inplace'):
            dst.copy_(src.to(dst.dtype).reshape(dst.shape))
        else:
            raise ValueError("unknown req %r" % (req,))

    def forward(self, is_train, req, in_data, out_data, aux):
        raise NotImplementedError()

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        raise NotImplementedError()


class CustomOpProp(object):
    def __init__(self, need_top_grad=False):
        self.need_top_grad_ = need_top_grad

    def list_arguments(self):
        return ['data']

    def list_outputs(self):
        return ['output']

    def list_auxiliary_states(self):
        return []

    def infer_shape(self, in_shape):
        return in_shape, [in_shape[0]], []

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return []

    def create_operator(self, ctx, shapes, dtypes):
        raise NotImplementedError()


def register(reg_name):
    def deco(prop_cls):
        _REGISTRY[reg_name] = prop_cls
        return prop_cls
    return deco


def get_prop(op_type):
    if op_type not in _REGISTRY:
        raise KeyError("custom op %r is not registered (have %s)" % (op_type, sorted(_REGISTRY)))
    return _REGISTRY[op_type]


def Custom(*args, **kwargs):
    """Eager counterpart of `mx.sym.Custom(op_type=..., **tensors_and_attrs)`: tensor kwargs
    are matched to list_arguments(), everything else is stringified (MXNet passes every
    attribute as a string) and handed to the Prop constructor.  Returns the outputs."""
    op_type = kwargs.pop('op_type')
    kwargs.pop('name', None)
    prop_cls = get_prop(op_type)
    tensors = {k: v for k, v in kwargs.items() if torch.is_tensor(v)}
    attrs = {k: str(v) for k, v in kwargs.items() if not torch.is_tensor(v)}
    prop = prop_cls(**attrs)
    names = prop.list_arguments()
    in_data = list(args) + [tensors[n] for n in names[len(args):]]
    in_shapes = [tuple(t.shape) for t in in_data]
    res = prop.infer_shape(in_shapes)
    out_shapes = res[1]
    dev = in_data[0].device
    out_data = [torch.empty(s, device=dev, dtype=torch.float32) for s in out_shapes]
    op = prop.create_operator(None, in_shapes, None)
    op.forward(False, ['write'] * len(out_data), in_data, out_data, [])
    return out_data[0] if len(out_data) == 1 else out_data


from . import proposal  # noqa: E402,F401  (registers "proposal")
from . import learn_nms  # noqa: E402,F401  (registers "learn_nms")
from . import targets  # noqa: E402,F401  (registers "proposal_target", "BoxAnnotatorOHEM", "nms_multi_target")
