"""Builds ablated copies of libse_hip.so (textual edits of csrc/se_gemm.hip, compiled to tools/micro/bin/libse_hack_<name>.so) to
find what bounds conv3_bf16_kernel: each variant removes one cost (results are WRONG, only the timing means anything).
Time them with tools/micro/run_hacks.sh <Cin> ... on the GPU box."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PKG = os.path.join(ROOT, 'speech-enhancement_amd')
SRC = open(os.path.join(PKG, 'csrc', 'se_gemm.hip')).read()
FLAGS = ['--offload-arch=gfx950', '-O3', '-munsafe-fp-atomics', '-fPIC', '-std=c++17', '-Wno-unused-result', '-Xclang',
         '-target-feature', '-Xclang', '-packed-fp32-ops', '-I', os.path.join(PKG, 'csrc')]


def kernel_region(s):
    a = s.index('void conv3_bf16_kernel(GemmArgs g) {')
    b = s.index('// Row-panel kernel for the token-wise layers')
    return a, b


def edit(s, old, new, count=1):
    a, b = kernel_region(s)
    body = s[a:b]
    assert body.count(old) >= 1, old
    return s[:a] + body.replace(old, new, count) + s[b:]


VARIANTS = {
    'nobar2': lambda s: edit(s, '''      __syncthreads();
    }
  }
  float* cs = reinterpret_cast<float*>(Ap) + wave * 32 * 36;''', '''    }
  }
  float* cs = reinterpret_cast<float*>(Ap) + wave * 32 * 36;'''),
    'nosplitA': lambda s: edit(s, 'for (int i = 0; i < 4; ++i) split_store<NPL>(ra[i], &Ap[(1 + r0 + 32 * i) * SA + kq * 4], PA);',
                               'for (int i = 0; i < 4; ++i) for (int q = 0; q < NPL; ++q) *reinterpret_cast<u32x2_*>(&Ap[q * PA + (1 + r0 + 32 * i) * SA + kq * 4]) = (u32x2_){__builtin_bit_cast(unsigned, ra[i].x), __builtin_bit_cast(unsigned, ra[i].y)};'),
    'noloadA': lambda s: edit(s, 'ra[i] = buf_load4_(Ar, (unsigned)q < (unsigned)Mb ? (unsigned)q * (unsigned)d.lda * 4u + cb : BUF_OOB_);',
                              'ra[i] = make_float4(1.f + q, 2.f, 3.f, cb);'),
    'nomask': lambda s: edit(s, 'if (df != 0) {', 'if (false) {'),
    # fragments from registers instead of LDS: all of them / only the weight fragments / only the activation fragments
    'nofrag': lambda s: edit(s, '''          af[q] = *reinterpret_cast<const bf16x8*>(&Ap[q * PA + ao]);
          bf0[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + bo]);
          bf1[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + 32 * SA + bo]);''', '''          { f32x4 z = {(float)(it + q), (float)ao, ra[0].x, 1.f}; af[q] = *reinterpret_cast<bf16x8*>(&z); z[0] += 1.f; bf0[q] = *reinterpret_cast<bf16x8*>(&z); z[1] += (float)bo; bf1[q] = *reinterpret_cast<bf16x8*>(&z); }'''),
    'nofragB': lambda s: edit(s, '''          bf0[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + bo]);
          bf1[q] = *reinterpret_cast<const bf16x8*>(&Bp[q * PB + 32 * SA + bo]);''', '''          { f32x4 z = {(float)(it + q), (float)bo, ra[0].x, 1.f}; bf0[q] = *reinterpret_cast<bf16x8*>(&z); z[1] += 1.f; bf1[q] = *reinterpret_cast<bf16x8*>(&z); }'''),
    'nofragA': lambda s: edit(s, '''          af[q] = *reinterpret_cast<const bf16x8*>(&Ap[q * PA + ao]);''', '''          { f32x4 z = {(float)(it + q), (float)ao, ra[0].x, 1.f}; af[q] = *reinterpret_cast<bf16x8*>(&z); }'''),
    'nostoreA': lambda s: edit(s, 'for (int i = 0; i < 4; ++i) split_store<NPL>(ra[i], &Ap[(1 + r0 + 32 * i) * SA + kq * 4], PA);',
                               'for (int i = 0; i < 4; ++i) asm volatile("" :: "v"(ra[i].x), "v"(ra[i].y), "v"(ra[i].z), "v"(ra[i].w));'),
    'nomfma': lambda s: edit(edit(s, 'acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf0[qb], acc0, 0, 0, 0);',
                                  'acc0[qa] += __builtin_bit_cast(f32x4, af[qa])[0] * __builtin_bit_cast(f32x4, bf0[qb])[1];'),
                             'acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[qa], bf1[qb], acc1, 0, 0, 0);',
                             'acc1[qa] += __builtin_bit_cast(f32x4, af[qa])[2] * __builtin_bit_cast(f32x4, bf1[qb])[3];'),
}


def build(name, text):
    os.makedirs(os.path.join(ROOT, 'tools', 'micro', 'bin'), exist_ok=True)
    src = f'/tmp/se_gemm_{name}.hip'
    open(src, 'w').write(text)
    obj = f'/tmp/se_gemm_{name}.o'
    subprocess.check_call(['/opt/rocm/bin/hipcc'] + FLAGS + ['-c', src, '-o', obj], stderr=subprocess.DEVNULL)
    objs = [os.path.join(PKG, 'build', f) for f in sorted(os.listdir(os.path.join(PKG, 'build'))) if f.endswith('.o') and
            f != 'se_gemm.hip.o']
    out = os.path.join(ROOT, 'tools', 'micro', 'bin', f'libse_hack_{name}.so')
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', out, obj] + objs)
    print('built', out)


if __name__ == '__main__':
    names = sys.argv[1:] or list(VARIANTS)
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(4) as ex:
        list(ex.map(lambda n: build(n, VARIANTS[n](SRC)), names))
