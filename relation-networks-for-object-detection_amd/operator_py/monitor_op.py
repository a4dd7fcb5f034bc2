"""`monitor` CustomOp of the reference (relation_rcnn/operator_py/monitor_op.py:15-55): an identity used to tap a tensor of
the graph while debugging (its call sites in symbols/ are commented out).  forward copies the input, backward passes the
gradient through; kept so that every name the reference registers resolves here too."""
from . import CustomOp, CustomOpProp, register, Custom


class MonitorOperator(CustomOp):
    def __init__(self, nickname):
        super(MonitorOperator, self).__init__()
        self.nickname = nickname

    def forward(self, is_train, req, in_data, out_data, aux):
        self.assign(out_data[0], req[0], in_data[0])

    def backward(self, req, out_grad, in_data, out_data, in_grad, aux):
        self.assign(in_grad[0], req[0], out_grad[0])


@register('monitor')
class MonitorProp(CustomOpProp):
    def __init__(self, nickname):
        super(MonitorProp, self).__init__(need_top_grad=False)
        self.nickname = nickname

    def list_arguments(self):
        return ['input']

    def list_outputs(self):
        return ['output']

    def infer_shape(self, in_shape):
        return [in_shape[0]], [in_shape[0]]

    def create_operator(self, ctx, shapes, dtypes):
        return MonitorOperator(self.nickname)

    def declare_backward_dependency(self, out_grad, in_data, out_data):
        return [out_grad[0]]


def monitor_wrapper(tensor, name):
    return Custom(input=tensor, op_type='monitor', nickname=name)
