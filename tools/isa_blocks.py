"""Per-basic-block instruction census of one compiled kernel: MFMA / other VALU / transcendental / DS read / DS write / VMEM /
SALU / waits, so that the issue budget of a loop body can be read off before a GPU run.
usage: python tools/isa_blocks.py <file.s> <mangled-name-regex> [min_instructions]"""
import re, sys

txt = open(sys.argv[1]).read()
pat = re.compile(sys.argv[2])
minin = int(sys.argv[3]) if len(sys.argv) > 3 else 40
parts = re.split(r'\n(_Z[\w]+):\s+; @', txt)
for i in range(1, len(parts), 2):
    name, body = parts[i], parts[i + 1]
    if not pat.search(name):
        continue
    meta = body[body.find('.Lfunc_end'):][:3000]
    info = {k: (re.findall(k + r': (\d+)', meta) or ['?'])[0] for k in ('NumVgprs', 'NumAgprs', 'TotalNumVgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize')}
    print(name, info)
    code = body.split('s_endpgm')[0].split('\n')
    blocks, cur, label = [], [], 'entry'
    for l in code:
        s = l.strip()
        if not s or s.startswith(';') or s.startswith('.'):
            if re.match(r'^\.LBB\d+_\d+:', s):
                blocks.append((label, cur)); cur = []; label = s.split(':')[0]
            continue
        cur.append(s)
    blocks.append((label, cur))
    tot = {}
    for label, ins in blocks:
        c = dict(mfma=0, valu=0, trans=0, dsr=0, dsw=0, vmem_ld=0, vmem_st=0, salu=0, wait=0, barrier=0, branch=0, dpp=0)
        for s in ins:
            op = s.split()[0]
            if op.startswith('v_mfma'): c['mfma'] += 1
            elif op.startswith(('v_exp', 'v_log', 'v_rcp', 'v_rsq', 'v_sqrt', 'v_sin', 'v_cos')): c['trans'] += 1
            elif op.startswith('v_'):
                c['valu'] += 1
                if 'dpp' in s or 'permlane' in op: c['dpp'] += 1
            elif op.startswith(('ds_read', 'ds_load', 'ds_bpermute', 'ds_permute', 'ds_swizzle')): c['dsr'] += 1
            elif op.startswith('ds_'): c['dsw'] += 1
            elif op.startswith(('global_load', 'buffer_load', 'flat_load', 'scratch_load')): c['vmem_ld'] += 1
            elif op.startswith(('global_store', 'buffer_store', 'flat_store', 'global_atomic', 'buffer_atomic', 'scratch_store')): c['vmem_st'] += 1
            elif op == 's_waitcnt': c['wait'] += 1
            elif op == 's_barrier': c['barrier'] += 1
            elif op.startswith(('s_cbranch', 's_branch')): c['branch'] += 1
            elif op.startswith('s_'): c['salu'] += 1
        for k, v in c.items():
            tot[k] = tot.get(k, 0) + v
        if len(ins) >= minin:
            print(f'  {label:12s} n={len(ins):5d} ' + ' '.join(f'{k}={v}' for k, v in c.items() if v))
    print('  TOTAL', tot)
