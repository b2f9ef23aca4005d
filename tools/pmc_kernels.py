"""HBM bytes per dispatch of every p2c_* kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB, FETCH_SIZE doubled
on gfx950 -- MI355X_MICROARCH.md "HBM"):   python tools/pmc_kernels.py FETCH_DIR WRITE_DIR > out.json"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from pmc_summary import summarise

rd, wr = summarise(sys.argv[1]), summarise(sys.argv[2])
out = {}
for k, v in rd.items():
    if 'p2c' not in k or 'FETCH_SIZE' not in v:
        continue
    w = wr.get(k, {}).get('WRITE_SIZE', 0.0)
    out[k] = {'dispatches': v['dispatches'], 'read_bytes': round(v['FETCH_SIZE'] * 2048.0), 'write_bytes': round(w * 1024.0),
              'bytes': round(v['FETCH_SIZE'] * 2048.0 + w * 1024.0)}
print(json.dumps(out, indent=1, sort_keys=True))
