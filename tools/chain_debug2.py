"""Debug aid: the shard-consistency property of tests/test_pose_head_gpu.py::test_full_size_properties, per variant."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib, ops
from oracle import pose_head as O
d = torch.device('cuda:0')
lib = _lib.lib()
lib.p2c_pose_head_set_time_parallel_max_batch(0)
spec = ops.PoseHeadSpec(kind='pose_changes_6d')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
gen = torch.Generator().manual_seed(B)
y = torch.randn(B, 16, 26, 6, generator=gen).to(d)
st = torch.randint(0, 4, (B,), generator=gen).int().to(d)
gt2 = torch.randn(B, 16, 26, 2, generator=gen).to(d)
gt2[torch.rand(B, 16, 26, generator=gen).to(d) < 0.1] = 0
gt3 = torch.randn(B, 16, 26, 3, generator=gen).to(d)
h = B // 2
for name, mb in (('joint', 1 << 30), ('chain', 0)):
    lib.p2c_pose_head_set_chain_min_batch(mb)
    l1, _ = ops.pose_head(y, spec, st, gt2d=gt2, gt3d=gt3)
    la, _ = ops.pose_head(y[:h], spec, st[:h], gt2d=gt2[:h], gt3d=gt3[:h])
    lb, _ = ops.pose_head(y[h:], spec, st[h:], gt2d=gt2[h:], gt3d=gt3[h:])
    lc, _ = ops.pose_head(y[h:].clone(), spec, st[h:].clone(), gt2d=gt2[h:].clone(), gt3d=gt3[h:].clone())
    print(name, 'full', l1.vector.tolist(), 'a', la.vector.tolist(), 'b', lb.vector.tolist(), 'b(clone)', lc.vector.tolist())
for sl, tag in ((slice(None), 'full'), (slice(0, h), 'a'), (slice(h, None), 'b')):
    o = O.pose_head(y[sl].double().cpu(), 'pose_changes_6d', st[sl].cpu(), gt2d=gt2[sl].double().cpu(), gt3d=gt3[sl].double().cpu())
    print('oracle', tag, [float(o[k]) for k in ('loc_2d', 'loc_3d', 'loc_2d_3d')])
