"""Summarise rocprofv3 --pmc result databases (p_results.db) per kernel: sums of each counter over the dispatches of a
kernel, divided by the dispatch count.  usage: python tools/pmc_summary.py out.json db1 [db2 ...]"""
import json, sqlite3, sys, collections, re

def tables(con):
    return [r[0] for r in con.execute("select name from sqlite_master where type='table'")]

def load(db):
    con = sqlite3.connect(db)
    tb = tables(con)
    def find(prefix):
        t = [x for x in tb if x.startswith(prefix)]
        return t[0] if t else None
    t_pmc, t_info, t_disp, t_sym = find('rocpd_pmc_event'), find('rocpd_info_pmc'), find('rocpd_kernel_dispatch'), find('rocpd_info_kernel_symbol')
    names = {r[0]: r[1] for r in con.execute(f'select id, name from {t_info}')}
    ksym = {r[0]: r[1] for r in con.execute(f'select id, kernel_name from {t_sym}')}
    disp = {r[0]: (ksym.get(r[1], str(r[1])), r[2], r[3]) for r in con.execute(f'select id, kernel_id, start, end from {t_disp}')}
    out = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    for ev, pmc, val in con.execute(f'select event_id, pmc_id, value from {t_pmc}'):
        # event_id -> dispatch id mapping: rocpd_kernel_dispatch.event_id
        pass
    ev2d = {r[0]: r[1] for r in con.execute(f'select event_id, id from {t_disp}')}
    for ev, pmc, val in con.execute(f'select event_id, pmc_id, value from {t_pmc}'):
        d = ev2d.get(ev)
        if d is None:
            continue
        k = disp[d][0]
        out[k][names[pmc]] += val
        cnt[k].add(d)
    for d, (k, s, e) in disp.items():
        dur[k] += (e - s)
    return out, cnt, dur

def short(k):
    k = re.sub(r'\(.*', '', k)
    return k[:90]

def main():
    res = collections.defaultdict(dict)
    for db in sys.argv[2:]:
        out, cnt, dur = load(db)
        for k in out:
            n = max(1, len(cnt[k]))
            r = res[short(k)]
            r['dispatches'] = n
            r['avg_us'] = round(dur[k] / n / 1e3, 1)
            for c, v in out[k].items():
                r[c] = v / n
    json.dump(res, open(sys.argv[1], 'w'), indent=1, sort_keys=True)
    for k, r in sorted(res.items(), key=lambda kv: -kv[1].get('avg_us', 0) * kv[1].get('dispatches', 1))[:12]:
        print(k, {a: (round(b, 1) if isinstance(b, float) else b) for a, b in r.items()})

main()
