"""per-step kernel table from a rocprofv3 kernel_stats.csv: python tools/show_stats.py <csv> [steps] [rows]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 7.0
nshow = int(sys.argv[3]) if len(sys.argv) > 3 else 70
tot = sum(int(r['TotalDurationNs']) for r in rows)
print(f'kernel time per step: {tot / steps / 1e6:.2f} ms')
acc = 0
for r in rows[:nshow]:
    acc += int(r['TotalDurationNs'])
    print(f"{r['Name'][:72]:72s} {int(r['Calls']) / steps:7.1f} {float(r['AverageNs']) / 1e3:9.1f} {int(r['TotalDurationNs']) / steps / 1e6:7.2f} {acc / tot * 100:5.1f}")
