"""Launch the pose-head forward / backward kernels a few times at one batch size (for rocprofv3 --pmc / --kernel-trace).

    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d OUT -- python3 tools/prof_kernels.py 8192
"""
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib, ops

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
kind = sys.argv[3] if len(sys.argv) > 3 else 'pose_changes_6d'
T, J = 16, 26
d = torch.device('cuda:0')
lib = _lib.lib()
g = torch.Generator(device=d).manual_seed(1)
ny = {'pose_changes_6d': (6,), 'pose_changes': (3, 3), 'absolute_loc': (3,)}[kind]
y = torch.randn((B, T, J) + ny, device=d, generator=g)
if kind == 'pose_changes_6d':
    y[..., 0] += 1.5
    y[..., 4] += 1.5
st = torch.randint(0, 4, (B,), device=d, generator=g).int()
gt2 = torch.randn(B, T, J, 2, device=d, generator=g)
gt3 = torch.randn(B, T, J, 3, device=d, generator=g)
spec = ops.PoseHeadSpec(kind=kind)
f32 = dict(dtype=torch.float32, device=d)
bufs = {'partials': torch.empty(lib.p2c_pose_head_workspace_floats(B), **f32), 'loss_sums': torch.empty(4, **f32),
        'losses': torch.empty(3, **f32), 'final_rel_rot': torch.empty(B, J, 3, 3, **f32)}
desc = ops._fill_desc(spec, y, st, None, None, gt2, gt3, bufs, {})
gl = torch.tensor([0.0, 0.0, 1.0], **f32)
gy = torch.empty_like(y)
s = torch.cuda.current_stream().cuda_stream
for _ in range(reps):
    _lib.check(lib.p2c_pose_head_fwd(ctypes.byref(desc), s), 'fwd')
    _lib.check(lib.p2c_pose_head_bwd(ctypes.byref(desc), _lib.grad_loss_pointers(vector=gl.data_ptr()), None, None, None, gy.data_ptr(), s), 'bwd')
torch.cuda.synchronize()
print('done', B, float(bufs['losses'][2]))
