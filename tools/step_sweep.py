"""Full train step (LinearAE pose_changes, loc_2d_3d, fp32) over batch sizes, both forms of the fused step's first launch:
per-launch device time through p2c_train_step_launch (HIP-graph timed, bench.fused_step_times) and the trainer's own step.
usage: python tools/step_sweep.py [B ...]   (P2C_SWEEP_FORMS=latency,stream)"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from pedestrians_video_2_carla_amd import _lib

os.environ['P2C_FUSED_TRAIN_MAX_B'] = str(1 << 20)
lib = _lib.lib()
dev = torch.device('cuda:0')
out = {}
forms = os.environ.get('P2C_SWEEP_FORMS', 'latency,stream').split(',')
for B in [int(a) for a in sys.argv[1:]] or [256, 512, 1024, 2048, 4096, 8192]:
    for form in forms:
        lib.p2c_train_step_set_stream_min_batch(1 if form == 'stream' else (1 << 30))
        flow, dm, trainer, batch = bench.build_step(dev, B, True, True)
        for i in range(10):
            trainer.train_step(flow, batch, i)
        torch.cuda.synchronize()
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            for i in range(100):
                trainer.train_step(flow, batch, i)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 100 * 1e6)
        res = bench.fused_step_times(dev, flow, trainer, batch)
        row = {'step_us': round(min(ts), 1), 'Mclips_s': round(B / min(ts), 2)}
        if res is not None:
            row.update({k: round(v, 1) for k, v in res[0].items()})
        out[f'B{B}_{form}'] = row
        print(B, form, row, flush=True)
        del flow, dm, trainer, batch
        torch.cuda.empty_cache()
print(json.dumps(out))
