"""Builds libse_hip.so (all HIP kernels + the C-ABI) for gfx950 with hipcc, in-tree.

Per translation unit: build/<name>.hip.o and build/<name>.hip.ru.txt -- the compiler's kernel-resource-usage remarks of the SAME
compile (registers, scratch, occupancy, LDS per kernel): tests/test_host.py::test_no_scratch_in_default_path_kernels reads them, so
a spill in a hot kernel fails the CPU suite instead of showing up as a slow step."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, 'csrc')
LIB = os.path.join(HERE, 'libse_hip.so')
OBJDIR = os.path.join(HERE, 'build')
HIPCC = os.environ.get('HIPCC', '/opt/rocm/bin/hipcc')
# packed fp32 VALU ops (v_pk_add/mul/fma_f32) are disabled: measured on gfx950 (tools/micro/issue_bench.hip) they execute on the
# matrix pipe -- each one stalls the MFMAs of every wave of the SIMD for ~6 cycles -- while two scalar fp32 ops overlap with them
BASE_FLAGS = ['--offload-arch=gfx950', '-O3', '-munsafe-fp-atomics', '-fPIC', '-std=c++17', '-Wno-unused-result',
              '-Rpass-analysis=kernel-resource-usage']
NO_PACKED = ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
# translation units without MFMAs whose inner loops are fp32 FMAs on channel pairs: v_pk_fma_f32 halves their VALU issue
PACKED_OK = {'se_dwconv.hip'}
FLAGS = BASE_FLAGS + NO_PACKED


def flags_for(src):
    return BASE_FLAGS if os.path.basename(src) in PACKED_OK else FLAGS


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.hip'))


def _headers():
    return [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith('.h')] + [os.path.join(HERE, '..', 'include', 'se_hip.h'),
                                                                                   os.path.abspath(__file__)]


def obj_of(src):
    return os.path.join(OBJDIR, os.path.basename(src) + '.o')


def ru_of(src):
    return os.path.join(OBJDIR, os.path.basename(src) + '.ru.txt')


def stale_sources():
    """translation units whose object (or resource-usage record) is older than the source, any header or this recipe"""
    ht = max(os.path.getmtime(h) for h in _headers())
    out = []
    for src in sources():
        o, r = obj_of(src), ru_of(src)
        if not (os.path.exists(o) and os.path.exists(r)) or os.path.getmtime(o) < max(os.path.getmtime(src), ht):
            out.append(src)
    return out


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return bool(stale_sources()) or any(os.path.getmtime(obj_of(s)) > t for s in sources())


def build(force=False, verbose=True):
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJDIR, exist_ok=True)
    todo = sources() if force else stale_sources()
    procs = []
    for src in todo:
        cmd = [HIPCC] + flags_for(src) + ['-c', src, '-o', obj_of(src)]
        if verbose:
            print(' '.join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
    for src, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            sys.stderr.write(err)
            raise RuntimeError(f'hipcc failed on {src}')
        remarks = [l for l in err.split('\n') if 'remark:' in l]
        other = [l for l in err.split('\n') if 'warning:' in l or 'error:' in l]
        if other and verbose:
            sys.stderr.write('\n'.join(other) + '\n')
        with open(ru_of(src), 'w') as f:
            f.write('\n'.join(remarks) + '\n')
    cmd = [HIPCC, '--offload-arch=gfx950', '-shared', '-fPIC', '-o', LIB] + [obj_of(s) for s in sources()]
    if verbose:
        print(' '.join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


def resource_usage():
    """[(translation unit, mangled kernel name, dict)] from the records of the last build: VGPRs, AGPRs, ScratchSize [bytes/lane],
    Occupancy [waves/SIMD], SGPRs, LDS Size [bytes/block]"""
    import re
    out = []
    for src in sources():
        if not os.path.exists(ru_of(src)):
            continue
        cur = None
        for line in open(ru_of(src)):
            m = re.search(r'remark: (?:\S+ )?\s*Function Name: (\S+)', line) or re.search(r'Function Name: (\S+)', line)
            if m:
                cur = {}
                out.append((os.path.basename(src), m.group(1), cur))
                continue
            m = re.search(r'remark:\s*(?:\[[^\]]*\]\s*)?\s*([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\d+)', line)
            if m and cur is not None:
                cur[m.group(1).strip()] = int(m.group(2))
    return out


if __name__ == '__main__':
    build(force='--force' in sys.argv)
