"""`python bench.py --gpus N` starts N ranks by itself (launch.py): dry run with 2 gloo ranks on CPU and a stub step."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra=None, timeout=240):
    env = dict(os.environ)
    for k in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT'):
        env.pop(k, None)
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + args, cwd=ROOT, env=env, timeout=timeout,
                          stdout=subprocess.PIPE, stderr=subprocess.PIPE, universal_newlines=True)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_bench_gpus2_spawns_two_ranks():
    r = _run(['--gpus', '2', '--stub', '--steps', '3', '--warmup', '1'])
    assert r.returncode == 0, r.stderr[-2000:]
    res = _json_line(r.stdout)                      # exactly ONE line: only rank 0 prints
    assert res['n_gpus'] == 2 and res['steps'] == 3 and res['warmup'] == 1
    assert res['config']['ranks_seen_by_collective'] == 2          # counted by an all-reduce over the ranks


def test_bench_single_rank_needs_no_launcher():
    r = _run(['--stub', '--steps', '2', '--warmup', '0'])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)['n_gpus'] == 1


def test_rank_count_mismatch_fails_loudly():
    """A torchrun environment with fewer ranks than --gpus must not print a line labelled with the requested count."""
    r = _run(['--gpus', '4', '--stub', '--steps', '1'], env_extra={'RANK': '0', 'WORLD_SIZE': '1', 'LOCAL_RANK': '0'})
    assert r.returncode != 0 and 'rank(s) came up' in (r.stderr + r.stdout)
    assert not [l for l in r.stdout.splitlines() if l.startswith('{')]


def test_torchrun_command_line():
    import relnet_amd  # noqa: F401
    from relnet_amd import launch
    cmd = launch.command('bench.py', ['--gpus', '8', '--steps', '5'], 8, port=29511)
    assert cmd[1:3] == ['-m', 'torch.distributed.run'] and '--nproc-per-node' in cmd and cmd[cmd.index('--nproc-per-node') + 1] == '8'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-5:] == ['bench.py', '--gpus', '8', '--steps', '5']
    os.environ.pop('RANK', None)
    assert not launch.under_torchrun()


def test_symbol_base_class_reports_every_bad_parameter():
    import numpy as np
    import relnet_amd  # noqa: F401
    from relnet_amd.utils_symbol import Symbol

    class G(object):
        def list_arguments(self): return ['data', 'w', 'b', 'label']
        def list_outputs(self): return ['out_output']
        def list_auxiliary_states(self): return ['mean']
        def infer_shape(self, **kw): return [kw['data'], (4, 3, 3, 3), (4,), (1,)], [(1, 4, 8, 8)], [(4,)]

    s = Symbol()
    s.sym = G()
    assert s.symbol is s.sym
    s.infer_shape({'data': (1, 3, 8, 8)})
    assert s.arg_shape_dict['w'] == (4, 3, 3, 3) and s.out_shape_dict['out_output'] == (1, 4, 8, 8) and s.aux_shape_dict['mean'] == (4,)
    assert abs(s.get_msra_std((4, 3, 3, 3)) - np.sqrt(2.0 / 27)) < 1e-12 and abs(s.get_msra_std((10, 5)) - np.sqrt(0.4)) < 1e-12
    ok = {'w': np.zeros((4, 3, 3, 3)), 'b': np.zeros(4)}
    s.check_parameter_shapes(ok, {'mean': np.zeros(4)}, {'data': (1, 3, 8, 8)}, is_train=False)       # 'label' skipped at test time
    with pytest.raises(AssertionError, match='label not initialized'):
        s.check_parameter_shapes(ok, {'mean': np.zeros(4)}, {'data': (1, 3, 8, 8)}, is_train=True)
    with pytest.raises(ValueError) as e:
        s.check_parameter_shapes({'w': np.zeros((4, 3, 1, 1))}, {}, {'data': (1, 3, 8, 8)}, is_train=False)
    msg = str(e.value)
    assert 'shape inconsistent for w' in msg and 'b not initialized' in msg and 'mean not initialized' in msg


def test_training_benchmark_lr_rule():
    """bench.bench_lr: the yaml's rate up to 16 summed images per step, then inversely proportional to the images summed over all
    ranks (the gradient is a SUM, train_end2end.py:167; random-init weights went non-finite at 4 ranks x 8 images at the yaml's rate)."""
    sys.path.insert(0, ROOT)
    import bench
    assert bench.bench_lr(0.0005, 1, 1) == 0.0005 and bench.bench_lr(0.0005, 8, 1) == 0.0005 and bench.bench_lr(0.0005, 16, 1) == 0.0005
    assert bench.bench_lr(0.0005, 8, 4) == pytest.approx(0.00025) and bench.bench_lr(0.0005, 8, 8) == pytest.approx(0.000125)
    assert bench.bench_lr(0.0005, 1, 8) == 0.0005
