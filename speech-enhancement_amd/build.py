"""Builds libse_hip.so (all HIP kernels + the C-ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libse_hip.so')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# packed fp32 VALU ops (v_pk_add/mul/fma_f32) are disabled: measured on gfx950 (tools/micro/issue_bench.hip) they execute on the
# matrix pipe -- each one stalls the MFMAs of every wave of the SIMD for ~6 cycles -- while two scalar fp32 ops overlap with them
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-munsafe-fp-atomics', '-fPIC', '-std=c++17', '-Wno-unused-result']
NO_PACKED = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
# translation units without MFMAs whose inner loops are fp32 FMAs on channel pairs: v_pk_fma_f32 halves their VALU issue
PACKED_OK = {'se_dwconv.hip'}
FLAGS = BASE_FLAGS + NO_PACKED


def flags_for(src):
    return BASE_FLAGS if os.path.basename(src) in PACKED_OK else FLAGS


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')]
    deps.append(os.path.join(HERE, '..', 'include', 'se_hip.h'))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    objs = []
    procs = []
    os.makedirs(os.path.join(HERE, 'build'), exist_ok=True)
    for src in sources():
        obj = os.path.join(HERE, 'build', os.path.basename(src) + '.o')
        objs.append(obj)
        cmd = [HIPCC] + flags_for(src) + ['-c', src, '-o', obj]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f'hipcc failed on {src}')
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + objs
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == '__main__':
    build(force='--force' in sys.argv)
