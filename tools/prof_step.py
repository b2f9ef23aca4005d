"""Run N captured train steps of the benchmark configuration (for rocprofv3 --kernel-trace / --pmc passes).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/prof_step.py 256 50
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
device = torch.device('cuda:0')
flow, dm, trainer, batch = bench.build_step(device, B, True, True)
for i in range(steps):
    loss = trainer.train_step(flow, batch, i)
torch.cuda.synchronize()
print('done', B, steps, float(loss))
