"""Build recipe of oracle/_ref (TEST INFRASTRUCTURE ONLY).

Compiles the reference's own CUDA sources for this path -- unedited, from where they lie under /root/reference -- with
hipcc for gfx950 into oracle/_ref/libref_cuda.so:

    relation_rcnn/operator_cxx/nn/deformable_im2col.cuh     deformable_im2col / col2im / col2im_coord kernels
    relation_rcnn/operator_cxx/deformable_psroi_pooling.cu  DeformablePSROIPool forward / backward kernels
    lib/nms/nms_kernel.cu                                   nms_kernel + host function _nms (the body behind gpu_nms.pyx)

behind oracle/refshim_cuda/ (stub MXNet / mshadow / dmlc headers, CUDA-runtime names mapped to HIP, a C driver).  The
reference's build system (MXNet's make with nvcc, setup.py with Cython + nvcc) is not run.  No reference source is copied
into this repository: the .so is git-ignored and rebuilt by this script wherever /root/reference exists; it travels to the
GPU box with the snapshot, where tests/test_gpu_refcuda.py and tests/golden/gen_golden_gpu.py load it.

Two libraries are built: `libref_cuda.so` with -ffp-contract=off (every product and sum of the kernels rounded
separately -- the arithmetic oracle/deform.py restates) and `libref_cuda_fma.so` with the compiler's default contraction
(nvcc's default is --fmad=true; WHICH multiply-adds it fuses cannot be known without nvcc, so this variant bounds the
effect instead of pinning it).

    python oracle/build_ref.py          # -> oracle/_ref/libref_cuda.so, oracle/_ref/libref_cuda_fma.so
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get('RELNET_REFERENCE', '/root/reference')
SHIM = os.path.join(HERE, 'refshim_cuda')
OUT = os.path.join(HERE, '_ref')
SOURCES = ['relation_rcnn/operator_cxx/nn/deformable_im2col.cuh', 'relation_rcnn/operator_cxx/deformable_psroi_pooling.cu',
           'lib/nms/nms_kernel.cu', 'lib/nms/gpu_nms.hpp']


def available():
    return all(os.path.exists(os.path.join(REF, s)) for s in SOURCES)


def up_to_date():
    """Both libraries exist and are newer than the shim sources and the reference files."""
    libs = [os.path.join(OUT, n) for n in ('libref_cuda.so', 'libref_cuda_fma.so')]
    if not all(os.path.exists(l) for l in libs):
        return False
    srcs = [os.path.join(REF, s) for s in SOURCES] + [os.path.join(SHIM, f) for f in ('refshim.h', 'ref_driver.hip')] + [os.path.abspath(__file__)]
    return min(os.path.getmtime(l) for l in libs) >= max(os.path.getmtime(f) for f in srcs)


def build(verbose=False):
    """-> list of built libraries ([] when /root/reference is absent: the GPU box uses the prebuilt files)."""
    if not available():
        return []
    os.makedirs(OUT, exist_ok=True)
    hipcc = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
    inc = ['-I', SHIM, '-I', os.path.join(SHIM, 'sys'), '-I', os.path.join(SHIM, 'inc', 'a', 'b'),
           '-I', os.path.join(REF, 'relation_rcnn', 'operator_cxx'), '-I', os.path.join(REF, 'lib', 'nms')]
    base = [hipcc, '--offload-arch=gfx950', '-O2', '-std=c++14', '-fPIC', '-x', 'hip', '-Wno-everything',
            '-fdelayed-template-parsing',     # bodies of the reference's never-instantiated host wrappers (`kernel << <grid, ... >> >(...)`, an nvcc-only spelling) stay unparsed
            '-DMXNET_OPERATOR_DEFORMABLE_PSROI_POOLING_INL_H_',      # skips deformable_psroi_pooling-inl.h (operator plumbing, not on the path)
            '-include', os.path.join(SHIM, 'refshim.h')] + inc
    libs = []
    for name, extra in (('libref_cuda.so', ['-ffp-contract=off']), ('libref_cuda_fma.so', [])):
        objs = []
        for src in (os.path.join(SHIM, 'ref_driver.hip'), os.path.join(REF, 'lib', 'nms', 'nms_kernel.cu')):
            obj = os.path.join(OUT, os.path.splitext(os.path.basename(src))[0] + '_' + name.replace('.so', '.o'))
            cmd = base + extra + ['-c', src, '-o', obj]
            if verbose:
                print(' '.join(cmd))
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=900)
            if r.returncode != 0:
                raise RuntimeError('oracle/_ref: hipcc failed on %s:\n%s' % (src, r.stdout.decode()[-4000:]))
            objs.append(obj)
        lib = os.path.join(OUT, name)
        r = subprocess.run([hipcc, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', lib] + objs, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, timeout=900)
        if r.returncode != 0:
            raise RuntimeError('oracle/_ref: link failed:\n%s' % r.stdout.decode()[-4000:])
        for o in objs:
            os.remove(o)
        libs.append(lib)
    return libs


if __name__ == '__main__':
    print(build(verbose='-v' in sys.argv))
