#!/bin/bash
# After `ONLY_STEP=1 bash tools/collect_profiles.sh gpurun_out/X` on the GPU box: copy the fused train step's summaries into
# profiles/r04 and replace its entries of profiles/traffic.json (the other entries keep their stamps).  usage: bash tools/refresh_step_profiles.sh gpurun_out/X
set -e
R=${1:?collection directory}
D=profiles/r04
for B in 256 1024 8192; do for C in FETCH_SIZE WRITE_SIZE; do python3 tools/pmc_summary.py $R/pmc_step_b${B}_$C > $D/pmc_step_b${B}_$C.json; done; done
python3 tools/pmc_summary.py $R/pmc_stream_b8192_SQ_INSTS > $D/pmc_stream_b8192_SQ_INSTS.json
python3 tools/pmc_summary.py $R/pmc_stream_b8192_SQ_CYCLES > $D/pmc_stream_b8192_SQ_CYCLES.json
cp $R/bench_default.json $D/bench_default.json
cp $(ls $R/bench_stats/*/*kernel_stats.csv | head -1) $D/bench_b256_noextra_kernel_stats.csv
cp $(ls $R/step_b1024_stats/*/*kernel_stats.csv | head -1) $D/step_b1024_kernel_stats.csv
cp $(ls $R/step_b8192_stats/*/*kernel_stats.csv | head -1) $D/step_b8192_kernel_stats.csv
python3 tools/make_traffic.py /tmp/traffic_step.json 256:$R/pmc_step_b256_FETCH_SIZE:$R/pmc_step_b256_WRITE_SIZE 1024:$R/pmc_step_b1024_FETCH_SIZE:$R/pmc_step_b1024_WRITE_SIZE 8192:$R/pmc_step_b8192_FETCH_SIZE:$R/pmc_step_b8192_WRITE_SIZE > /dev/null
python3 - <<'PY'
import json, sys
sys.path.insert(0, 'tools')
from traffic_stamp import src_sha16
t = json.load(open('profiles/traffic.json'))
n = json.load(open('/tmp/traffic_step.json'))
for k, v in n.items():
    if k.startswith('train_'):
        t[k] = v
json.dump(t, open('profiles/traffic.json', 'w'), indent=1, sort_keys=True)
print('stale entries:', [k for k, v in t.items() if v.get('src_sha16') != src_sha16(k)])
PY
