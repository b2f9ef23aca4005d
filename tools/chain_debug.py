"""Debug aid: chain-lane vs joint-lane pose-head kernels on one random lean call; per-joint error table."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pedestrians_video_2_carla_amd import _lib, ops
from oracle import pose_head as O

def main():
    B, T = int(sys.argv[1]) if len(sys.argv) > 1 else 8, int(sys.argv[2]) if len(sys.argv) > 2 else 4
    transform = sys.argv[3] if len(sys.argv) > 3 else 'hips_neck_bbox'
    up = tuple(float(v) for v in sys.argv[4].split(',')) if len(sys.argv) > 4 else (0.0, 0.0, 1.0)
    d = torch.device('cuda:0')
    lib = _lib.lib()
    lib.p2c_pose_head_set_time_parallel_max_batch(0)
    gen = torch.Generator().manual_seed(5)
    y = torch.randn(B, T, 26, 6, generator=gen); y[..., 0] += 1.5; y[..., 4] += 1.5
    st = torch.randint(0, 4, (B,), generator=gen)
    tgt = O.synthetic_batch(B, T, seed=6, missing_prob=0.0)
    gt2 = tgt['projection_2d_transformed'] if transform != 'none' else tgt['projection_2d']
    gt3 = tgt['absolute_pose_loc']
    spec = ops.PoseHeadSpec(kind='pose_changes_6d', transform=transform)
    res = {}
    for name, mb in (('joint', 1 << 30), ('chain', 0)):
        lib.p2c_pose_head_set_chain_min_batch(mb)
        yd = y.to(d).requires_grad_(True)
        losses, _ = ops.pose_head(yd, spec, st.to(d).int(), gt2d=gt2.to(d), gt3d=gt3.to(d))
        w = torch.tensor(up, device=d)
        (losses.vector * w).sum().backward()
        res[name] = (losses.vector.detach().cpu(), yd.grad.detach().cpu())
    y64 = y.double().requires_grad_(True)
    o = O.pose_head(y64, 'pose_changes_6d', st, gt2d=gt2.double(), gt3d=gt3.double(), transform=transform)
    (up[0] * o['loc_2d'] + up[1] * o['loc_3d'] + up[2] * o['loc_2d_3d']).backward()
    print('losses joint', res['joint'][0].tolist(), 'chain', res['chain'][0].tolist(), 'oracle', [float(o[k]) for k in ('loc_2d', 'loc_3d', 'loc_2d_3d')])
    g = y64.grad
    sc = float(g.abs().max())
    for name in ('joint', 'chain'):
        e = (res[name][1].double() - g).abs()
        print(name, 'max err', float(e.max()) / sc)
        pj = e.amax(dim=(0, 1, 3)) / sc
        print('  per joint:', ' '.join(f'{v:.1e}' for v in pj.tolist()))
        pt = e.amax(dim=(0, 2, 3)) / sc
        print('  per frame:', ' '.join(f'{v:.1e}' for v in pt.tolist()))
        pb = e.amax(dim=(1, 2, 3)) / sc
        print('  per clip :', ' '.join(f'{v:.1e}' for v in pb.tolist()))

main()
