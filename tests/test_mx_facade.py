"""The `mx` graph-builder facade (graph-builder row) on the CPU: graph construction rules, JSON round
trip, shape inference -- and, when the reference checkout is present (build container), that the reference's own
graph files, imported UNCHANGED, build all ten shipped experiments in both modes and reproduce the committed
fixtures of tests/golden/symbols/ (which the GPU tests execute)."""
import glob
import json
import os
import subprocess
import sys

import pytest

import relnet_amd  # noqa: F401
from relnet_amd import mx, config as C

HERE = os.path.dirname(os.path.abspath(__file__))
SYM_DIR = os.path.join(HERE, 'golden', 'symbols')
REF = '/root/reference'


def _load(name):
    return mx.sym.load(os.path.join(SYM_DIR, name + '.json'))


def test_composition_rules():
    mx.sym.reset_names()
    data = mx.sym.Variable('data')
    fc = mx.sym.FullyConnected(name='fc_new_1', data=data, num_hidden=1024)
    assert fc.list_arguments() == ['data', 'fc_new_1_weight', 'fc_new_1_bias'] and fc.list_outputs() == ['fc_new_1_output']
    bn = mx.symbol.BatchNorm(name='bn1', data=mx.symbol.Convolution(name='c1', data=data, num_filter=8, kernel=(3, 3), no_bias=True),
                             use_global_stats=True, fix_gamma=False, eps=1e-5)
    assert bn.list_arguments() == ['data', 'c1_weight', 'bn1_gamma', 'bn1_beta']
    assert bn.list_auxiliary_states() == ['bn1_moving_mean', 'bn1_moving_var']
    a = mx.sym.Activation(data=bn, act_type='relu')              # unnamed operators are numbered per operator type
    b = mx.sym.Activation(data=a, act_type='relu')
    assert (a.name, b.name) == ('activation0', 'activation1')
    s = mx.sym.SoftmaxOutput(name='cls_prob', data=fc, normalization='valid')
    assert 'cls_prob_label' in s.list_arguments()
    parts = mx.sym.split(data=data, num_outputs=4, axis=1)
    x0, x1, x2, x3 = parts                                       # tuple unpacking of multi-output symbols
    assert len(parts) == 4 and x2.list_outputs() == [parts.list_outputs()[2]]
    e = (1.0 / 8.0) * (x0 + x1) - x2 / 2
    ops = [n.op for n in e._topo() if n.op != 'null']
    assert ops == ['split', '_plus', '_mul_scalar', '_div_scalar', '_minus']
    g = mx.sym.Group([fc, mx.sym.BlockGrad(e)])
    assert len(g.list_outputs()) == 2
    arg, out, aux = bn.infer_shape(data=(2, 3, 10, 12))
    assert arg == [(2, 3, 10, 12), (8, 3, 3, 3), (8,), (8,)] and out == [(2, 8, 8, 10)] and aux == [(8,), (8,)]
    r = mx.sym.Reshape(mx.sym.Variable('p'), shape=(-3, -2))
    assert r.infer_shape(p=(5, 7, 64))[1] == [(35, 64)]
    assert mx.sym.Reshape(mx.sym.Variable('p'), shape=(0, 2, -1, 0)).infer_shape(p=(1, 24, 38, 63))[1] == [(1, 2, 456, 63)]
    m = mx.sym.maximum(left=r, right=1e-6)
    assert m._topo()[-1].op == '_maximum_scalar'
    back = mx.sym.load_json(g.tojson())
    assert back.list_arguments() == g.list_arguments() and back.list_outputs() == g.list_outputs()
    assert json.loads(back.tojson()) == json.loads(g.tojson())


@pytest.mark.parametrize('name,n_args,n_aux,outs', [
    ('rcnn_end2end_8epoch_test', 330, 208, {'rois_output': (300, 5), 'cls_prob_reshape_output': (1, 300, 81), 'bbox_pred_reshape_output': (1, 300, 8)}),
    ('rcnn_end2end_relation_8epoch_test', 346, 208, {'rois_output': (300, 5), 'cls_prob_reshape_output': (1, 300, 81)}),
    ('rcnn_end2end_relation_learn_nms_8epoch_test', 360, 208, {'rois_output': (300, 5), 'learn_nms_sorted_bbox': (100, 80, 4)}),
])
def test_fixture_graphs_infer_shapes(name, n_args, n_aux, outs):
    sym = _load(name)
    assert len(sym.list_arguments()) == n_args and len(sym.list_auxiliary_states()) == n_aux
    arg, out, aux = sym.infer_shape(data=(1, 3, 600, 1000), im_info=(1, 3))
    shapes = dict(zip(sym.list_arguments(), arg))
    assert shapes['fc_new_1_weight'] == (1024, 12544) and shapes['res4b22_branch2c_weight'] == (1024, 256, 1, 1)
    assert shapes['rpn_cls_score_weight'] == (24, 512, 1, 1) and shapes['conv1_weight'] == (64, 3, 7, 7)
    o = dict(zip(sym.list_outputs(), out))
    for k, v in outs.items():
        assert o[k] == v, (k, o)
    assert all(s is not None for s in arg + aux)
    assert json.loads(mx.sym.load_json(sym.tojson()).tojson()) == json.loads(sym.tojson())


def test_relation_graph_parameter_names_match_detector():
    """Every argument of the reference's relation test graph is a key of the Detector's parameter dict (same names)."""
    from relnet_amd import backbone
    sym = _load('rcnn_end2end_relation_8epoch_test')
    p = backbone.init_params(seed=1)
    need = [a for a in sym.list_arguments() if a not in ('data', 'im_info')] + sym.list_auxiliary_states()
    assert not [n for n in need if n not in p]
    arg, _, aux = sym.infer_shape(data=(1, 3, 600, 1000), im_info=(1, 3))
    for n, s in list(zip(sym.list_arguments(), arg)) + list(zip(sym.list_auxiliary_states(), aux)):
        if n in p:
            assert tuple(p[n].shape) == s, (n, tuple(p[n].shape), s)


def test_experiment_presets_have_their_symbols():
    for name, (symbol, _) in C.EXPERIMENTS.items():
        cfg = C.experiment(name)
        assert cfg.symbol == symbol and cfg.dataset.NUM_CLASSES == 81 and cfg.network.NUM_ANCHORS == 12
        import pickle
        back = pickle.loads(pickle.dumps(cfg))                   # SYM_REL:220 pickles the tree into proposal_target
        assert back.TRAIN.BBOX_STDS == cfg.TRAIN.BBOX_STDS and back['TEST']['RPN_POST_NMS_TOP_N'] == 300
    with pytest.raises(ValueError):
        C._overlay(C._defaults(), {'NOT_A_KEY': 1})


@pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference checkout (build container only)')
def test_reference_graph_files_run_unchanged_and_match_the_fixtures():
    r = subprocess.run([sys.executable, os.path.join(HERE, 'golden', 'gen_symbol_json.py'), '--ref', REF, '--check'],
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=os.path.dirname(HERE))
    assert r.returncode == 0, r.stdout.decode()[-2000:]
    assert r.stdout.decode().count('same') == len(glob.glob(os.path.join(SYM_DIR, '*.json')))


@pytest.mark.skipif(not os.path.isdir(REF), reason='needs the reference checkout (build container only)')
def test_all_ten_reference_experiments_build_in_both_modes():
    code = r'''
import importlib, os, sys
sys.path.insert(0, %r)
import relnet_amd
from relnet_amd import mx, config as C, py2compat
mx.install(); py2compat.add_source_dir(os.path.join(%r, 'relation_rcnn', 'symbols'))
for name in C.EXPERIMENTS:
    cfg = C.experiment(name)
    update = C.update_config(os.path.join(%r, 'experiments/relation_rcnn/cfgs/resnet_v1_101_coco_trainvalminus_' + name + '.yaml'), C._defaults())
    assert update.symbol == cfg.symbol
    net = getattr(importlib.import_module(cfg.symbol), cfg.symbol)()
    for is_train in (False, True):
        sym = net.get_symbol_rcnn(cfg, is_train=is_train) if 'fpn' in name else net.get_symbol(cfg, is_train=is_train)
        assert len(sym.list_arguments()) > 300 and len(sym.list_outputs()) >= 3
        if not is_train and 'fpn' not in name:
            net.infer_shape({'data': (1, 3, 600, 1000), 'im_info': (1, 3)})
            assert net.out_shape_dict['rois_output'] == (300, 5)
    print('built', name)
''' % (os.path.dirname(HERE), REF, REF)
    r = subprocess.run([sys.executable, '-c', code], stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    assert r.returncode == 0, r.stdout.decode()[-3000:]
    assert r.stdout.decode().count('built') == 10


def test_operator_corner_cases_of_the_registry():
    """pow(scalar, Symbol) / scalar ** Symbol is scalar ** x (an `_rpower_scalar` node), tile left-pads `reps` with 1 like MXNet,
    and Reshape(reverse=True) keeps a -4 split together with its two arguments when the code list is reversed."""
    import torch
    from relnet_amd.mx import registry as R
    mx.sym.reset_names()
    x = mx.sym.Variable('x')
    assert mx.sym.pow(2.0, x)._topo()[-1].op == '_rpower_scalar' and (2.0 ** x)._topo()[-1].op == '_rpower_scalar'
    assert mx.sym.pow(x, 2.0)._topo()[-1].op == '_power_scalar' and (x ** 2.0)._topo()[-1].op == '_power_scalar'
    assert mx.sym.pow(2.0, x).infer_shape(x=(3, 4))[1] == [(3, 4)]
    t = torch.tensor([[1.0, 2.0], [3.0, 0.5]])
    assert torch.allclose(R.OPS['_rpower_scalar'].fn({'scalar': '2.0'}, t), 2.0 ** t)
    assert mx.sym.tile(x, reps=(2,)).infer_shape(x=(3, 4))[1] == [(3, 8)]          # reps (2,) == (1, 2)
    assert mx.sym.tile(x, reps=(2, 1, 3)).infer_shape(x=(3, 4))[1] == [(2, 3, 12)]
    # MXNet doc example: shape (10, 5, 4), Reshape(shape=(-1, 0), reverse=True) -> (50, 4); forward would give (40, 5)
    assert R.reshape_codes((10, 5, 4), (-1, 0), reverse=True) == (50, 4) and R.reshape_codes((10, 5, 4), (-1, 0)) == (40, 5)
    # -4 under reverse: the LAST axis is split, its factors keep their order
    assert R.reshape_codes((6, 20), (0, -4, 4, 5), reverse=True) == (6, 4, 5)
    assert R.reshape_codes((6, 20), (0, -4, -1, 5), reverse=True) == (6, 4, 5)
    assert R.reshape_codes((20, 6), (-4, 4, 5, 0)) == (4, 5, 6)
