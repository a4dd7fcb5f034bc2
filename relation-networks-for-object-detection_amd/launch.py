"""One process per GPU from ONE command: `python bench.py --gpus N` (or any driver script) re-executes itself under
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free>` when it was
not started by torchrun already.

The reference starts its data-parallel job from one command as well: `train_end2end.py:69-71` builds `ctx = [mx.gpu(i) for
i in config.gpus.split(',')]` and `core/module.py` / `DataParallelExecutorGroup.py` drive one executor per device from a
single process.  Here every device gets its own process (torch.distributed over RCCL / xGMI), so the command has to spawn
them; a job that comes up with fewer ranks than requested fails loudly instead of reporting a smaller run under the
requested name.
"""
import json
import os
import socket
import subprocess
import sys
import time


def under_torchrun():
    """True when RANK / WORLD_SIZE were provided by a launcher (torchrun, the round driver, mpirun wrappers)."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    return port


def command(script, argv, nproc, port=None):
    """The torchrun command line for `nproc` local ranks of `script argv` (rendezvous on 127.0.0.1: container host names
    may not resolve)."""
    return [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(int(nproc)),
            '--master-addr', '127.0.0.1', '--master-port', str(port or free_port()), script] + list(argv)


def respawn(script, argv, nproc, env=None):
    """Run `script argv` as `nproc` ranks and return the job's exit code (the caller should `sys.exit` with it).
    stdout / stderr are inherited: rank 0 prints the result line."""
    e = dict(os.environ if env is None else env)
    e.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL fails with hipIpcGetMemHandle errors otherwise
    e.setdefault('OMP_NUM_THREADS', '8')
    return subprocess.call(command(script, argv, nproc), env=e)


def stub_bench(a):
    """Dry run of the launcher + measurement protocol without GPUs: `a.gpus` gloo ranks, a small CPU matmul as the
    step, the same barrier / max-over-ranks timing and JSON fields as the real bench (tests/test_launch.py)."""
    import torch
    from . import dist as D
    rank, world, _ = D.init(backend='gloo')
    if world != a.gpus:
        raise SystemExit("bench.py: --gpus %d but %d rank(s) came up (WORLD_SIZE)" % (a.gpus, world))
    seen = int(D.sum_over_ranks(1))
    x = torch.randn(64, 64)
    for _ in range(a.warmup):
        x = (x @ x).tanh()
    D.fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        x = (x @ x).tanh()
    D.fence()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    if rank == 0:
        print(json.dumps({'metric': 'stub steps/s', 'value': world * a.steps / elapsed, 'unit': 'steps/s', 'n_gpus': world,
                          'steps': a.steps, 'warmup': a.warmup, 'ms_per_step': 1e3 * elapsed / a.steps,
                          'higher_is_better': True, 'scaling': 'weak', 'data': 'stub',
                          'config': {'workload': 'launcher dry run (gloo, CPU)', 'ranks_seen_by_collective': seen}}))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    return 0
