"""rocprofv3 target: a few train steps at batch B (default 8192) through the throughput forms of the fused step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
os.environ['P2C_FUSED_TRAIN_MAX_B'] = str(1 << 20)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
flow, dm, trainer, batch = bench.build_step(torch.device('cuda:0'), B, True, True)
for i in range(30):
    trainer.train_step(flow, batch, i)
torch.cuda.synchronize()
