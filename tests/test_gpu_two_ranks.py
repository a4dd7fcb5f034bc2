"""The N > 1 training exchange on a one-GPU box: two ranks share cuda:0 and exchange over gloo (tools/dist_check.py).
With the same batch on both ranks the bucketed all-reduce must deliver world x the local gradients -- for the eager step (buckets
announced from inside the backward pass, heads -> res5 -> res4 (second half) -> res4 (first half) -> res3) and for train.CapturedStep (buckets between hipGraph
segments).  Everything but the collective backend (gloo instead of RCCL) is the code path of `bench.py --gpus N --train`."""
import ast
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
pytestmark = pytest.mark.gpu


def test_two_ranks_on_one_device_sum_their_gradients():
    from importlib import import_module
    launch = import_module('relation-networks-for-object-detection_amd.launch')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes', '1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(launch.free_port()), os.path.join(ROOT, 'tools', 'dist_check.py')]
    r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
    out = r.stdout.decode()
    assert r.returncode == 0, r.stderr.decode()[-3000:]
    line = [l for l in out.splitlines() if l.startswith('DIST_CHECK')]
    assert line, out[-2000:]
    res = ast.literal_eval(line[-1][len('DIST_CHECK'):].strip())
    assert res['finite']
    # five buckets in buffer order res3 | res4 (units a .. b10) | res4 (b11 .. b22) | res5 | heads: announced last to first
    assert res['eager_order'] == [4, 3, 2, 1, 0] and res['captured_segments'][:5] == [4, 3, 2, 1, 0]
    # overlapped protocol (what bench.py times): nothing is waited for before update(), which takes the buckets in launch order
    assert res['overlap_launch_order'] == [4, 3, 2, 1, 0] and res['update_order'] == [4, 3, 2, 1, 0], res
    assert res['weights_moved'] and res['weights_equal_across_ranks'], res
    # atomically accumulated fp32 gradients: the order of the adds differs between two passes
    assert res['eager_max_rel_diff'] < 1e-3 and res['captured_max_rel_diff'] < 1e-3, res
    assert res['eager_bias_exact']
