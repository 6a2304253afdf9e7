"""Static scan of the compiled kernels for the things that cost this project the most without showing in the source:
scratch use, occupancy 1, global / buffer loads issued right behind a full `s_waitcnt vmcnt(0)` (serialised loads: usually a
spilled address or an exec-masked region per load), and stores / atomics behind one (serialised stores: a flag-dependent load
or a guarded prefetch the wait-count pass has to assume pending).  usage: python tools/isa_scan.py   (compiles every csrc/*.hip with -save-temps)"""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CS = os.path.join(ROOT, 'speech-enhancement_amd', 'csrc')
tmp = tempfile.mkdtemp()
rows = []
for src in sorted(glob.glob(CS + '/*.hip')):
    base = os.path.basename(src)[:-4]
    fl = [] if base == 'se_dwconv' else ['-Xclang', '-target-feature', '-Xclang', '-packed-fp32-ops']
    subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-munsafe-fp-atomics', '-fPIC', '-std=c++17', '-Wno-unused-result'] + fl +
                   ['-c', src, '-o', os.path.join(tmp, base + '.o'), '-save-temps=obj'], cwd=tmp, stderr=subprocess.DEVNULL)
    txt = open(os.path.join(tmp, base + '-hip-amdgcn-amd-amdhsa-gfx950.s')).read()
    parts = re.split(r'\n(_Z[\w]+):\s+; @', txt)
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1]
        meta = body[:body.find('; Occupancy') + 40]
        g = lambda k: int((re.findall(k + r': (\d+)', meta) or ['0'])[0])
        code = [l.strip() for l in body.split('s_endpgm')[0].split('\n') if l.strip() and not l.strip().startswith(';')]
        ser = sum(1 for j, l in enumerate(code) if (l.startswith('global_load') or l.startswith('buffer_load')) and
                  any(p.startswith('s_waitcnt vmcnt(0)') for p in code[max(0, j - 4):j]))
        # loads whose result is waited for at once (a full vmcnt(0) within two instructions): exposed latency unless other waves cover it
        imm = sum(1 for j, l in enumerate(code) if (l.startswith('global_load') or l.startswith('buffer_load')) and
                  any(p.startswith('s_waitcnt vmcnt(0)') for p in code[j + 1:j + 3]))
        # stores / atomics issued right behind a full vmcnt(0): each then also waits for the store before it (vmcnt retires in order)
        sts = sum(1 for j, l in enumerate(code) if (l.startswith('global_store') or l.startswith('buffer_store') or l.startswith('global_atomic')) and
                  any(p.startswith('s_waitcnt') and 'vmcnt(0)' in p for p in code[max(0, j - 6):j]))
        if g('ScratchSize') or g('Occupancy') <= 1 or ser >= 4 or sts >= 8 or (imm >= 4 and '--imm' in sys.argv):
            rows.append((base, name, g('TotalNumVgprs'), g('ScratchSize'), g('Occupancy'), ser, imm, sts))
names = subprocess.run(['c++filt'], input='\n'.join(r[1] for r in rows), capture_output=True, text=True).stdout.strip().split('\n')
for r, d in zip(rows, names):
    print(f'{r[0]:10s} {d[:84]:84s} vgpr {r[2]:3d} scratch {r[3]:4d} occ {r[4]} loads-behind-vmcnt0 {r[5]} waited-at-once {r[6]} stores-behind-vmcnt0 {r[7]}')
