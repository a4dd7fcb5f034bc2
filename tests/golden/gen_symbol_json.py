#!/usr/bin/env python
"""Build the reference's graphs by running ITS OWN graph files (relation_rcnn/symbols/*.py, imported unchanged from
/root/reference) on the `relnet_amd.mx` facade, and store them as JSON fixtures under tests/golden/symbols/.

Build container only (the GPU box has no reference checkout): the `-m gpu` tests load these fixtures with
`mx.sym.load`, bind them on the GPU and compare with the hand-wired Detector and the oracle.

    python tests/golden/gen_symbol_json.py [--ref /root/reference] [--check]
"""
import argparse
import importlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

#: (experiment, is_train, entry) -> fixture name
FIXTURES = [('rcnn_end2end_8epoch', False), ('rcnn_end2end_relation_8epoch', False), ('rcnn_end2end_relation_8epoch', True),
            ('rcnn_end2end_relation_learn_nms_8epoch', False), ('rcnn_dcn_end2end_relation_8epoch', False),
            ('rcnn_fpn_relation_8epoch', False)]


def build(ref, name, is_train):
    import relnet_amd  # noqa: F401
    from relnet_amd import mx, config as C, py2compat
    mx.install()
    py2compat.add_source_dir(os.path.join(ref, 'relation_rcnn', 'symbols'))
    cfg = C.experiment(name)
    mx.sym.reset_names()
    mod = importlib.import_module(cfg.symbol)
    net = getattr(mod, cfg.symbol)()
    sym = net.get_symbol_rcnn(cfg, is_train=is_train) if 'fpn' in name else net.get_symbol(cfg, is_train=is_train)
    return sym


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--check', action='store_true', help='compare with the committed fixtures instead of writing')
    a = ap.parse_args()
    out = os.path.join(HERE, 'symbols')
    os.makedirs(out, exist_ok=True)
    for name, is_train in FIXTURES:
        sym = build(a.ref, name, is_train)
        text = json.dumps(json.loads(sym.tojson()), separators=(',', ':'))
        path = os.path.join(out, '%s_%s.json' % (name, 'train' if is_train else 'test'))
        if a.check:
            assert open(path).read() == text, path
            print('same', path)
        else:
            with open(path, 'w') as f:
                f.write(text)
            print('wrote %s (%d bytes, %d nodes)' % (path, len(text), len(json.loads(text)['nodes'])))


if __name__ == '__main__':
    main()
