"""Lists, per kernel, the `s_waitcnt vmcnt(0)` that sit INSIDE a loop within a few instructions behind a load issued in the same
loop iteration -- the signature of a prefetch that is waited for at once (e.g. a load under `if`: the join copies the loaded
registers, DESIGN_APPENDIX A00).  usage: python tools/isa_loop_waits.py file.s [--steady] [name filter ...]
(--steady: only waits behind the kernel's first s_barrier, i.e. not the one-time staging loops in front of a persistent tile loop);
tests/test_host.py::test_no_defeated_prefetch_in_persistent_kernels calls scan() on the round-5 kernels."""
import re, subprocess, sys


def scan(path, steady=False):
    """{demangled kernel name: [(line, distance to the last load)]}"""
    txt = open(path).read()
    parts = re.split(r'\n(_Z[\w]+):\s+; @', txt)
    out = []
    for i in range(1, len(parts), 2):
        name, body = parts[i], parts[i + 1].split('s_endpgm')[0]
        lines = body.split('\n')
        hits = []
        since_load = None
        seen_barrier = False
        for j, l in enumerate(lines):
            t = l.strip()
            if t.startswith('s_barrier'):
                seen_barrier = True
            if t.startswith(('buffer_load', 'global_load')):
                since_load = j
            if t.startswith('s_waitcnt') and 'vmcnt(0)' in t and since_load is not None and j - since_load <= 25:
                inloop = any('in Loop' in x or 'This Inner Loop' in x or 'This Loop' in x for x in lines[max(0, j - 60):j]
                             if x.strip().startswith(('.LBB', ';')))
                if inloop and (seen_barrier or not steady):
                    hits.append((j, j - since_load))
        if hits:
            out.append((name, hits))
    names = subprocess.run(['c++filt'], input='\n'.join(n for n, _ in out), capture_output=True, text=True).stdout.strip().split('\n')
    return {d: h for (n, h), d in zip(out, names)}


if __name__ == '__main__':
    args = [a for a in sys.argv[1:] if a != '--steady']
    for d, h in scan(args[0], steady='--steady' in sys.argv).items():
        if args[1:] and not any(f in d for f in args[1:]):
            continue
        print(f'{d[:110]:110s} {len(h)} waits: ' + ' '.join(f'L{j}(+{k})' for j, k in h[:8]))
