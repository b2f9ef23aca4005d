"""List the kernels of one train step from a rocprofv3 kernel trace (CSV): start offset, duration, gap, grid, name.
usage: python tools/step_trace.py <kernel_trace.csv> [marker substring = adamw]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else 'adamw'
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if marker in r['Kernel_Name']]
a, b = idx[-3], idx[-2]
step = rows[a + 1:b + 1]
t0 = int(step[0]['Start_Timestamp'])
prev = t0
busy = 0
for r in step:
    st, en = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].replace('void ', '').replace('at::native::', '')[:78]
    print(f"{(st - t0) / 1e3:8.1f} +{(en - st) / 1e3:6.1f} gap{(st - prev) / 1e3:5.1f} g={r['Grid_Size_X']:>7} {name}")
    prev = en
    busy += en - st
print(len(step), 'kernels,', (int(step[-1]['End_Timestamp']) - t0) / 1e3, 'us span,', busy / 1e3, 'us busy')
