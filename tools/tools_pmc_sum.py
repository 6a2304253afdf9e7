import csv, collections, sys
pat = sys.argv[2] if len(sys.argv) > 2 else ''
tot = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][:48]
    if pat in k:
        tot[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
for k, v in tot.items():
    print(k)
    for c, x in sorted(v.items()): print(f'    {c:28s} {x / n[(k, c)]:.4g}')
