"""Base class the reference's model classes derive from (`from utils.symbol import Symbol`, lib/utils/symbol.py:10-56).

The interface (attribute and method names, what each method leaves behind) is fixed by the unchanged symbol files that
subclass it; the implementation is this repository's: shape tables are built by one helper over the facade graph's three
name lists, and a shape mismatch raises a `ValueError` that lists EVERY offending parameter instead of stopping at the
first one.
"""
import math


class ParameterShapeError(AssertionError, ValueError):
    """What check_parameter_shapes raises; an AssertionError like the reference's `assert`s, so callers that catch those keep working."""


def _shape_table(names, shapes):
    return {n: tuple(int(d) for d in s) for n, s in zip(names, shapes)}


class Symbol(object):
    #: graph argument names that are inputs rather than parameters when testing (labels are absent at test time)
    _LABEL_TAG = 'label'

    def __init__(self):
        self.sym = None
        self.arg_shape_dict = self.out_shape_dict = self.aux_shape_dict = None

    symbol = property(lambda self: self.sym)

    # -- provided by the model classes in relation_rcnn/symbols/*.py ---------------------------------------------
    def get_symbol(self, cfg, is_train=True):
        raise NotImplementedError("model classes build their graph here and store it in self.sym")

    def init_weights(self, cfg, arg_params, aux_params):
        raise NotImplementedError("model classes fill arg_params / aux_params here")

    # -- helpers the model classes call ---------------------------------------------------------------------------
    @staticmethod
    def get_msra_std(shape):
        """He-normal standard deviation sqrt(2 / fan_in) of a weight [out, in, *kernel]."""
        return math.sqrt(2.0 / (float(shape[1]) * math.prod(int(d) for d in shape[2:])))

    def infer_shape(self, data_shape_dict):
        graph = self.sym
        args, outs, auxs = graph.infer_shape(**data_shape_dict)
        self.arg_shape_dict = _shape_table(graph.list_arguments(), args)
        self.out_shape_dict = _shape_table(graph.list_outputs(), outs)
        self.aux_shape_dict = _shape_table(graph.list_auxiliary_states(), auxs)

    def check_parameter_shapes(self, arg_params, aux_params, data_shape_dict, is_train=True):
        """Every graph parameter must be present with the inferred shape (inputs, and labels at test time, are skipped)."""
        def is_input(name):
            return name in data_shape_dict or (not is_train and self._LABEL_TAG in name)

        problems = []
        todo = [(n, arg_params, self.arg_shape_dict) for n in self.sym.list_arguments() if not is_input(n)]
        todo += [(n, aux_params, self.aux_shape_dict) for n in self.sym.list_auxiliary_states()]
        for name, given, inferred in todo:
            if name not in given:
                problems.append('%s not initialized' % name)
            elif tuple(given[name].shape) != tuple(inferred[name]):
                problems.append('shape inconsistent for %s inferred %s provided %s'
                                % (name, tuple(inferred[name]), tuple(given[name].shape)))
        if problems:
            raise ParameterShapeError('; '.join(problems))
