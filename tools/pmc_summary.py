"""Summarise rocprofv3 --pmc csv output per kernel: mean counter value per dispatch (skipping the first dispatch)."""
import csv
import glob
import json
import sys
from collections import defaultdict


def summarise(directory):
    out = defaultdict(lambda: defaultdict(list))
    for path in glob.glob(directory + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(path)):
            out[r['Kernel_Name'].split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    res = {}
    for k, cs in out.items():
        res[k] = {c: sum(v[1:]) / max(1, len(v[1:])) if len(v) > 1 else v[0] for c, v in cs.items()}
        res[k]['dispatches'] = len(next(iter(cs.values())))
    return res


if __name__ == '__main__':
    print(json.dumps(summarise(sys.argv[1]), indent=1))
