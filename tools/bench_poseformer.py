"""cfg5 timing: pose_lifting flow, PoseFormer (own restatement), 81-frame clips, B=32 per GPU -- captured train steps.
usage: python tools/bench_poseformer.py [B=32] [steps=10] [bf16]"""
import json
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
from pedestrians_video_2_carla_amd.modules.flow.pose_lifting import LitPoseLiftingFlow
from pedestrians_video_2_carla_amd.modules.movements.pose_former import PoseFormer
from pedestrians_video_2_carla_amd.trainer import Trainer, seed_everything

B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
bf16 = 'bf16' in sys.argv[3:]
share = 'share' in sys.argv[3:]
d = torch.device('cuda:0')
seed_everything(22742)
dm = SyntheticCarlaRecordedDataModule(clip_length=81, batch_size=B)
kw = dict(compute_dtype=torch.bfloat16) if bf16 else {}
if share:
    kw['share_spatial'] = True
model = PoseFormer(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, clip_length=81, **kw)
flow = LitPoseLiftingFlow(movements_model=model, loss_modes=['loc_2d_3d'], transform='hips_neck_bbox')
trainer = Trainer(device=d, use_graph=True).setup(flow, dm)
batch = dm.generate_batch(d)
for i in range(3):
    trainer.train_step(flow, batch, i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = trainer.train_step(flow, batch, i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({'config': 'pose_lifting PoseFormer clip 81', 'B': B, 'bf16': bf16, 'share_spatial': share, 'ms_per_step': round(dt * 1e3, 3),
                  'clips_per_s': round(B / dt, 1), 'loss': float(loss)}))
