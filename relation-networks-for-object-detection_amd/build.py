"""Build librelnet_hip.so (the C-ABI of include/relnet_hip.h) from csrc/*.hip for gfx950.

hipcc cross-compiles without a GPU; the .so is kept in-tree (git-ignored) so that it
travels to the GPU box with the repo snapshot.
"""
import fcntl
import glob
import hashlib
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'librelnet_hip.so')
STAMP = os.path.join(HERE, 'csrc', '.build_stamp')
GUARD = os.path.join(HERE, 'csrc', '.asm_guard.json')        # report of the ISA guard below (written by the build that linked the library)
FLAGS = ['--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-gpu-rdc', '-Wno-unused-value',
         '-munsafe-fp-atomics']      # float atomicAdd = global_atomic_add_f32 (gradient buffers are ordinary device memory), not a CAS loop


def _digest():
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, '*.hip')) + glob.glob(os.path.join(CSRC, '*.h'))):
        h.update(os.path.basename(f).encode()); h.update(open(f, 'rb').read())     # file NAME, not its absolute path: the
        # library linked in this container is then reused as it is on the GPU box (same image, same hipcc)
    h.update(' '.join(FLAGS).encode())
    return h.hexdigest()


LAST_ACTION = None        # 'compiled' | 'reused' after build(): lets the driver see whether hipcc actually ran


def build(force=False, verbose=False):
    """Compile every csrc/*.hip and link the shared library; returns its path."""
    global LAST_ACTION
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        LAST_ACTION = 'reused'          # sources, flags and compiler unchanged since the library was linked
        return LIB
    # one builder at a time: the N ranks of `bench.py --gpus N` all call build(); the first one compiles, the others find the
    # stamp when they get the lock
    with open(os.path.join(CSRC, '.build_lock'), 'w') as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
            LAST_ACTION = 'reused'
            return LIB
        return _compile(dig, verbose)


def asm_agpr_guard(asm_path):
    """Scan the device assembly of gemm.hip: in every gemm_ring_kernel instantiation with SCHED 5 / 6 (the kernels whose
    accumulators are literal AGPRs kept across asm statements) collect the instructions outside `;;#ASMSTART .. ;;#ASMEND` that
    read or write an AGPR.  -> {'kernels_checked': n, 'asm_blocks': n, 'offenders': {kernel: [lines]}}"""
    import re
    cur, inasm = None, False
    checked, blocks, off = 0, 0, {}
    is_asm_kernel = re.compile(r'gemm_ring_kernelILi\d+ELi\d+ELi\d+ELi\d+E[tf]Li\dELi\d+ELi\d+ELb[01]ELi[56]E')
    agpr = re.compile(r'accvgpr|\ba\[\d|\ba\d+\b')
    with open(asm_path) as f:
        for ln in f:
            m = re.match(r'^(_ZN6relnet\S+):', ln)
            if m:
                cur = m.group(1) if is_asm_kernel.search(m.group(1)) else None
                inasm = False
                checked += cur is not None
                continue
            if cur is None:
                continue
            if ';;#ASMSTART' in ln:
                inasm = True; blocks += 1
            elif ';;#ASMEND' in ln:
                inasm = False
            elif ln.lstrip().startswith('s_endpgm'):
                cur = None
            elif not inasm and agpr.search(ln.split(';')[0]):
                off.setdefault(cur, []).append(ln.strip())
    return {'kernels_checked': checked, 'asm_blocks': blocks, 'offenders': off}


def _compile(dig, verbose):
    global LAST_ACTION
    LAST_ACTION = 'compiled'
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    objs = []
    procs = []
    hdr = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(CSRC, '*.h'))):
        hdr.update(open(f, 'rb').read())
    hdr.update(' '.join(FLAGS).encode())
    fresh = {}
    for src in sorted(glob.glob(os.path.join(CSRC, '*.hip'))):
        obj = os.path.splitext(src)[0] + '.o'
        objs.append(obj)
        # per-object stamp (source + headers + flags): an edit of one kernel file recompiles that file only
        od = hashlib.sha256(hdr.digest() + open(src, 'rb').read()).hexdigest()
        ostamp = obj + '.stamp'
        if os.path.exists(obj) and os.path.exists(ostamp) and open(ostamp).read() == od:
            continue
        fresh[obj] = od
        cmd = [hipcc] + FLAGS + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    # ISA guard of the hand-scheduled k-loops (gemm.hip, SCHED 5 / 6): their accumulators live in a[0:127] ACROSS separate asm
    # statements and the compiler only sees clobbers -- nothing in the language stops a future hipcc from parking a spill or an
    # AV-class temporary in those AGPRs between two slabs.  So the device assembly of gemm.hip is produced beside the object
    # (same flags, in parallel with the real compile) and checked: inside those kernels no instruction OUTSIDE the asm blocks may
    # touch an AGPR.  A violation fails the build instead of corrupting convolutions on some other toolchain.
    import json
    gemm_obj = os.path.join(CSRC, 'gemm.o')
    need_guard = gemm_obj in fresh or not os.path.exists(GUARD)       # (gemm.hip unchanged: the report of the build that compiled it stands)
    guard = None
    if need_guard:
        guard_s = os.path.join('/tmp', 'relnet_gemm_guard_%d.s' % os.getpid())
        guard = subprocess.Popen([hipcc] + FLAGS + ['--cuda-device-only', '-S', os.path.join(CSRC, 'gemm.hip'), '-o', guard_s],
                                 stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed on %s:\n%s" % (src, out.decode()))
    for obj, od in fresh.items():
        with open(obj + '.stamp', 'w') as f:
            f.write(od)
    if guard is not None:
        gout, _ = guard.communicate()
        if guard.returncode != 0:
            raise RuntimeError("hipcc -S failed on gemm.hip (ISA guard):\n%s" % gout.decode())
        report = asm_agpr_guard(guard_s)
        os.remove(guard_s)
    else:
        report = {k: v for k, v in json.load(open(GUARD)).items() if k != 'digest'}
    with open(GUARD, 'w') as f:
        json.dump(dict(report, digest=dig), f)
    if report['offenders']:
        raise RuntimeError("ISA guard: AGPR use outside the asm blocks of the hand-scheduled GEMM kernels (accumulators in a[0:127] "
                           "would be corrupted): %s" % json.dumps(report['offenders'])[:2000])
    tmp = LIB + '.tmp%d' % os.getpid()
    cmd = [hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', tmp] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n%s" % r.stdout.decode())
    os.replace(tmp, LIB)              # a process that has the old library mapped keeps its inode
    with open(STAMP, 'w') as f:
        f.write(dig)
    return LIB


if __name__ == '__main__':
    print(build(force=True, verbose=True))
