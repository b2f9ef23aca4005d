"""Debug aid: one case of test_chain_lane_kernels_lean_forward_and_backward, both mappings vs the fp64 / fp32 oracle."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
from pedestrians_video_2_carla_amd import _lib, ops
from oracle import pose_head as O
B, T, transform, kind = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
up = tuple(float(v) for v in sys.argv[5].split(','))
d = torch.device('cuda:0')
lib = _lib.lib()
lib.p2c_pose_head_set_time_parallel_max_batch(0)
gen = torch.Generator().manual_seed(B * 1000 + T)
y = torch.randn(B, T, 26, 6, generator=gen); y[..., 0] += 1.5; y[..., 4] += 1.5
st = torch.randint(0, 4, (B,), generator=gen)
tgt = O.synthetic_batch(B, T, seed=B * 1000 + T + 1, missing_prob=0.0)
gt2 = tgt['projection_2d_transformed'].clone()
gt2[torch.rand(B, T, 26, generator=gen) < 0.1] = 0.0
gt3, gt2_px = tgt['absolute_pose_loc'], tgt['projection_2d']
if transform == 'none': gt2 = gt2_px
elif transform != 'hips_neck_bbox': gt2 = O.normalize(gt2_px.double(), transform)[0].float()
if len(sys.argv) > 6:
    c = int(sys.argv[6]); y, st, gt2, gt3 = y[c:c + 1], st[c:c + 1], gt2[c:c + 1], gt3[c:c + 1]; B = 1
sl = (1, T - 1) if T > 4 else (0, T)
spec = ops.PoseHeadSpec(kind=kind, transform=transform, eval_slice=sl)
def orc(dt):
    yy = y.to(dt).clone().requires_grad_(True)
    o = O.pose_head(yy, kind, st, gt2d=gt2.to(dt), gt3d=gt3.to(dt), transform=transform, eval_slice=slice(*sl))
    (up[0] * o['loc_2d'] + up[1] * o['loc_3d'] + up[2] * o['loc_2d_3d']).backward()
    return o, yy.grad
o64, g64 = orc(torch.float64)
o32, g32 = orc(torch.float32)
sc = float(g64.abs().max())
e32 = (g32.double() - g64).abs()
print('oracle64 losses', [float(o64[k]) for k in ('loc_2d','loc_3d')], 'oracle32', [float(o32[k]) for k in ('loc_2d','loc_3d')], 'hn scale64', o64['projection_2d_scale'][0].tolist() if B == 1 else '');print('scale', sc, 'fp32 oracle max err', float(e32.max()), 'at clip', int(e32.amax(dim=(1, 2, 3)).argmax()))
for name, mb in (('joint', 1 << 30), ('chain', 0)):
    lib.p2c_pose_head_set_chain_min_batch(mb)
    from test_pose_head_gpu import run_hip
    losses, _, grad = run_hip(y, spec, st, gt2d=gt2, gt3d=gt3, want=(), upstream=up)
    e = (grad.cpu().double() - g64).abs()
    pc = e.amax(dim=(1, 2, 3))
    top = pc.topk(min(4, B))
    print(name, 'losses', losses.vector.tolist(), 'max err', float(e.max()), 'worst clips', top.indices.tolist(), [f'{v:.2e}' for v in top.values.tolist()],
          'fp32-oracle err there', [f'{float(e32[i].max()):.2e}' for i in top.indices.tolist()])
    i = int(top.indices[0])
    print('   worst clip per joint:', ' '.join(f'{v:.1e}' for v in e[i].amax(dim=(0, 2)).tolist()))
    print('   worst clip per frame:', ' '.join(f'{v:.1e}' for v in e[i].amax(dim=(1, 2)).tolist()))
