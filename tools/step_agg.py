"""Aggregate the kernels of the last full step of a rocprofv3 kernel trace by name.
usage: python tools/step_agg.py <kernel_trace.csv> <marker substring of a once-per-step kernel> [top=30]"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
marker = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-2], idx[-1]
step = rows[a:b]
agg = collections.defaultdict(lambda: [0, 0.0])
for r in step:
    n = r['Kernel_Name'][:110]
    agg[n][0] += 1
    agg[n][1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print('kernels', len(step), 'busy us', round(tot, 1), 'span us', (int(step[-1]['End_Timestamp']) - int(step[0]['Start_Timestamp'])) / 1e3)
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f"{t:10.1f}us {c:5d}x {t / c:9.1f}  {n}")
