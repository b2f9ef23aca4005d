"""cfg3 timing: autoencoder flow, Seq2SeqEmbeddings(pose_2d), B=512, T=16 -- eager train steps (MIOpen LSTM + HIP loss)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
from pedestrians_video_2_carla_amd.data.carla.carla_recorded_synthetic import SyntheticCarlaRecordedDataModule
from pedestrians_video_2_carla_amd.modules.flow.autoencoder import LitAutoencoderFlow
from pedestrians_video_2_carla_amd.modules.flow.output_types import MovementsModelOutputType as MT
from pedestrians_video_2_carla_amd.modules.movements.seq2seq import Seq2SeqEmbeddings
from pedestrians_video_2_carla_amd.trainer import Trainer, seed_everything

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
use_graph = len(sys.argv) > 3 and sys.argv[3] == 'graph'
if os.environ.get('P2C_BLAS'):
    torch.backends.cuda.preferred_blas_library(os.environ['P2C_BLAS'])
d = torch.device('cuda:0')
seed_everything(22742)
dm = SyntheticCarlaRecordedDataModule(clip_length=16, batch_size=B)
model = Seq2SeqEmbeddings(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON, movements_output_type=MT.pose_2d)
if os.environ.get('P2C_NO_FOLD'):
    model.fold_embeddings = False
flow = LitAutoencoderFlow(movements_model=model, loss_modes=['loc_2d'], transform='hips_neck_bbox')
trainer = Trainer(device=d, use_graph=use_graph).setup(flow, dm)
batch = dm.generate_batch(d)
first = []
for i in range(5):
    first.append(float(trainer.train_step(flow, batch, i)))
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    loss = trainer.train_step(flow, batch, i)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(json.dumps({'config': 'autoencoder Seq2SeqEmbeddings pose_2d', 'B': B, 'hip_graph': use_graph, 'ms_per_step': round(dt * 1e3, 3),
                  'clips_per_s': round(B / dt, 1), 'loss': float(loss), 'first_losses': first}))
