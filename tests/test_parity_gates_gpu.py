"""GPU: the parity gates of BASELINE.json's north_star at the benchmark configuration, and anchors that do not come from
oracle/ ("Tighten the parity gates").

  * 200-step loss curve at B = 256 with the reference's DEFAULT init: GPU fp32 and CPU fp32 are each compared with the fp64
    oracle curve; the GPU may deviate by max(1e-4, 2 x what CPU fp32 itself deviates) at every step -- the rule
    tests/test_pose_head_gpu.py uses for ill-conditioned gradients (with the default init the 6-D outputs start near zero
    and a1/|a1| amplifies rounding in ANY fp32 implementation; the conditioned B = 32 run in tests/test_flow_gpu.py stays
    the strict 1e-4 gate).
  * cfg5's pose head: absolute_loc at T = 81, eval_slice (4, 77), BODY_25 targets, vs the oracle.
  * hand-computed known answers for rotation_6d_to_matrix (Zhou et al. 2019, rows b1, b2, b1 x b2) and the screen-space
    camera (u = 400 + 400 x0 / (3.1 - x1), v = 300 + 400 (x2 + 1.2) / (3.1 - x1), w = 1 / (3.1 - x1))
    fed through the HIP kernel: these two third-party leaves are otherwise parity-unpinned.
"""
import math
import os
import sys

import pytest
import torch

from oracle import pose_head as O

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from test_flow_gpu import close, dev, make  # noqa: E402


def _cpu_curve(flow, batch, steps, dtype):
    from cpu_backend import StubDataModule, oracle_backend
    from pedestrians_video_2_carla_amd.trainer import Trainer
    flow = flow.to(dtype)
    tc = Trainer().setup(flow, StubDataModule())
    cb = (batch[0].to('cpu', dtype), {k: v.to('cpu', dtype) for k, v in batch[1].items()},
          {'age': batch[2]['age'], 'gender': batch[2]['gender']})
    with oracle_backend():
        return torch.stack([tc.train_step(flow, cb, i).double() for i in range(steps)])


def test_loss_curve_at_the_benchmark_configuration_default_init():
    from pedestrians_video_2_carla_amd.trainer import Trainer
    steps, B = 200, 256
    d = dev()
    flow_g, dm = make(B=B, missing=0.1)                # seed_everything(22742) inside: identical default init each time
    flow_32, _ = make(B=B, missing=0.1)
    flow_64, _ = make(B=B, missing=0.1)
    batch = dm.generate_batch(d)
    tg = Trainer(device=d, use_graph=True).setup(flow_g, dm)
    gpu = torch.stack([tg.train_step(flow_g, batch, i).clone() for i in range(steps)]).double().cpu()
    assert getattr(flow_g, '_pair_counts', None) is not None, 'the benchmark configuration takes the two-launch step'
    ref64 = _cpu_curve(flow_64, batch, steps, torch.float64)
    cpu32 = _cpu_curve(flow_32, batch, steps, torch.float32)
    dev_gpu = (gpu - ref64).abs() / ref64.abs()
    dev_cpu = (cpu32 - ref64).abs() / ref64.abs()
    # a deviation, once there, is carried forward by the optimisation: the allowance at step i is what CPU fp32 has shown
    # up to step i
    allowed = torch.maximum(torch.full_like(dev_cpu, 1e-4), 2.0 * torch.cummax(dev_cpu, 0).values)
    worst = int((dev_gpu / allowed).argmax())
    print(f'loss curve B={B} default init: {float(gpu[0]):.4f} -> {float(gpu[-1]):.4f}; max rel dev vs fp64: GPU '
          f'{float(dev_gpu.max()):.2e}, CPU fp32 {float(dev_cpu.max()):.2e}; tightest step {worst}: GPU '
          f'{float(dev_gpu[worst]):.2e} vs allowed {float(allowed[worst]):.2e}')
    assert torch.isfinite(gpu).all() and gpu[-1] < gpu[0]
    assert (dev_gpu <= allowed).all(), f'step {worst}: GPU {float(dev_gpu[worst]):.3e} > allowed {float(allowed[worst]):.3e}'
    # the first step has no history: it must meet the plain 1e-4 gate
    assert dev_gpu[0] <= 1e-4


@pytest.mark.parametrize('B,steps', [(256, 200), (1024, 200)])
def test_loss_curve_at_the_benchmark_configuration_conditioned_init_strict(B, steps):
    """The strict gate of BASELINE.json's north_star ("loss curve matching CPU reference to 1e-4") AT the benchmark
    configuration: B = 256, T = 16, 10 % missing input joints, 200 optimizer steps of the two-launch step (K13, replayed as its
    recorded call), against the CPU pipeline (LinearAE fp32 + fp32 oracle + AdamW) step by step, 1e-4 relative with no
    allowance. The init is the conditioned one of tests/test_flow_gpu.py (last-layer bias = identity rotations: with the
    default init a1/|a1| of near-zero 6-D outputs makes ANY fp32 run chaotic -- that run is the smoke test above); the fp64
    curve is printed beside it."""
    from pedestrians_video_2_carla_amd.trainer import Trainer
    d = dev()      # (B = 1024 = cfg2 / cfg4's per-GPU batch: the throughput form of the first launch, csrc/p2c_train_stream.hip)

    def conditioned():
        flow, dm = make(B=B, missing=0.1)
        last = flow.movements_model._LinearAE__decoder[4]
        with torch.no_grad():
            last.bias.copy_(torch.tensor([1., 0., 0., 0., 1., 0.]).repeat(26))
        return flow, dm

    flow_g, dm = conditioned()
    flow_32, _ = conditioned()
    flow_64, _ = conditioned()
    batch = dm.generate_batch(d)
    assert float((batch[0] == 0).all(-1).float().mean()) > 0.05, 'the missing-joint deformation is on'
    tg = Trainer(device=d, use_graph=True).setup(flow_g, dm)
    gpu = torch.stack([tg.train_step(flow_g, batch, i).clone() for i in range(steps)]).double().cpu()
    assert getattr(flow_g, '_pair_counts', None) is not None and tg._direct is not None, 'the two-launch step, direct replay'
    cpu32 = _cpu_curve(flow_32, batch, steps, torch.float32)
    ref64 = _cpu_curve(flow_64, batch, steps, torch.float64)
    dev_32 = ((gpu - cpu32).abs() / cpu32.abs())
    dev_64 = ((gpu - ref64).abs() / ref64.abs())
    cpu_64 = ((cpu32 - ref64).abs() / ref64.abs())
    print(f'strict curve B={B} conditioned init, 10% missing: {float(gpu[0]):.4f} -> {float(gpu[-1]):.4f}; max rel dev GPU vs '
          f'CPU fp32 {float(dev_32.max()):.2e} (step {int(dev_32.argmax())}), GPU vs fp64 {float(dev_64.max()):.2e}, CPU fp32 vs '
          f'fp64 {float(cpu_64.max()):.2e}')
    assert torch.isfinite(gpu).all() and gpu[-1] < 0.5 * gpu[0]
    assert float(dev_32.max()) < 1e-4, f'GPU vs CPU fp32: {float(dev_32.max()):.3e} at step {int(dev_32.argmax())}'
    assert float(dev_64.max()) < 1e-4, f'GPU vs fp64: {float(dev_64.max()):.3e} at step {int(dev_64.argmax())}'


def test_cfg5_pose_head_absolute_loc_t81_body25_targets():
    """PoseFormer's head (BASELINE.json configs[4]): absolute_loc output, clip_length 81, eval_slice (4, 77), targets in the
    BODY_25 layout (21 common joints) -- losses and grad_y vs the oracle."""
    from pedestrians_video_2_carla_amd import ops
    from pedestrians_video_2_carla_amd.data.base.skeleton import get_common_indices
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON
    B, T = 6, 81
    g = torch.Generator().manual_seed(81)
    st = torch.randint(0, 4, (B,), generator=g)
    base = O.absolute_tensors(torch.float64)[0][st]                            # (B, 26, 3) reference poses
    y = (base[:, None] + 0.05 * torch.randn(B, T, 26, 3, generator=g, dtype=torch.float64)).float()
    out_idx, in_idx = get_common_indices(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    gt2 = torch.zeros(B, T, 25, 3)
    gt3 = torch.zeros(B, T, 25, 3)
    gt2[:, :, in_idx, :2] = torch.randn(B, T, len(in_idx), 2, generator=g)
    gt2[..., 2] = 0.9                                                          # OpenPose confidence channel
    gt2[torch.rand(B, T, 25, generator=g) < 0.1] = 0.0                         # undetected joints
    gt3[:, :, in_idx] = y[:, :, out_idx] + 0.02 * torch.randn(B, T, len(in_idx), 3, generator=g)
    hips_col = in_idx.index(BODY_25_SKELETON.MidHip.value)
    gm = ops.joint_maps(out_idx, in_idx, 25)
    spec = ops.PoseHeadSpec(kind='absolute_loc', gmap2d=gm, gmap3d=gm, hips_lane=out_idx[hips_col], eval_slice=(4, 77))
    d = dev()
    yd = y.to(d).requires_grad_(True)
    losses, _ = ops.pose_head(yd, spec, st.to(d).int(), gt2d=gt2.to(d), gt3d=gt3.to(d))
    losses[2].backward()
    refs = {}
    for dt in (torch.float64, torch.float32):
        yr = y.to(dt).clone().requires_grad_(True)
        o = O.pose_head(yr, 'absolute_loc', st, gt2d=gt2.to(dt), gt3d=gt3.to(dt), out_idx=out_idx, in_idx=in_idx,
                        hips_col=hips_col, eval_slice=slice(4, 77))
        o['loc_2d_3d'].backward()
        refs[dt] = (o, yr.grad)
    o, gref = refs[torch.float64]
    for i, k in enumerate(('loc_2d', 'loc_3d', 'loc_2d_3d')):
        close(losses[i], o[k], k)
    err32 = (refs[torch.float32][1].double() - gref).abs().max().item() / gref.abs().max().item()
    close(yd.grad, gref, 'grad_y', rtol=max(1e-4, 2 * err32))
    assert (yd.grad[:, :4] == 0).all() and (yd.grad[:, 77:] == 0).all()        # frames outside the eval slice carry no loss


def test_rotation_6d_and_camera_known_answers():
    """Known answers worked out by hand (not by oracle/): 6-D -> R of three vectors through the relative_rot_6d kind, and the
    projection of the reference skeleton's hips / neck (SURVEY appendix A.5 numbers) through the pinhole camera."""
    from pedestrians_video_2_carla_amd import ops
    d = dev()
    cases = [
        ((1., 0., 0., 0., 1., 0.), ((1., 0., 0.), (0., 1., 0.), (0., 0., 1.))),
        # a1 = (0,2,0) -> b1 = (0,1,0); a2 = (1,1,0) -> a2 - (b1.a2) b1 = (1,0,0) = b2; b3 = b1 x b2 = (0,0,-1)
        ((0., 2., 0., 1., 1., 0.), ((0., 1., 0.), (1., 0., 0.), (0., 0., -1.))),
        # a1 = (3,0,4) -> b1 = (.6,0,.8); a2 = (0,5,0) is already orthogonal -> b2 = (0,1,0); b3 = b1 x b2 = (-.8,0,.6)
        ((3., 0., 4., 0., 5., 0.), ((.6, 0., .8), (0., 1., 0.), (-.8, 0., .6))),
    ]
    y = torch.zeros(len(cases), 1, 26, 6)
    y[..., 0] = 1.0
    y[..., 4] = 1.0                                                            # identity everywhere ...
    for i, (d6, _) in enumerate(cases):
        y[i, 0, 5] = torch.tensor(d6)                                          # ... except joint 5 (crl_arm__L)
    st = torch.zeros(len(cases), dtype=torch.int32)                            # adult female
    spec = ops.PoseHeadSpec(kind='relative_rot_6d', transform='none')
    _, outs = ops.pose_head(y.to(d), spec, st.to(d), want=('relative_pose_rot', 'projection_2d', 'absolute_pose_loc'))
    for i, (_, R) in enumerate(cases):
        got = outs['relative_pose_rot'][i, 0, 5].cpu()
        assert torch.allclose(got, torch.tensor(R), atol=2e-7), (i, got)
    # camera: clip 0 is the identity pose of the adult-female skeleton. Hips sit at the origin: (400, 300 + 480/3.1, 1/3.1);
    # neck at (0, -0.056796, -0.463826) (A.5): u = 400, v = 300 + 400 (1.2 - 0.463826) / (3.1 + 0.056796) = 393.2815
    # identity relative rotations are NOT the reference pose (its bones are rotated), so take the kernel's own abs loc and
    # re-derive the projection by hand from it for every joint
    x = outs['absolute_pose_loc'][0, 0].double().cpu()
    p = outs['projection_2d'][0, 0].double().cpu()
    Z = 3.1 - x[:, 1]
    want = torch.stack((400.0 + 400.0 * x[:, 0] / Z, 300.0 + 400.0 * (x[:, 2] + 1.2) / Z, 1.0 / Z), -1)
    assert torch.allclose(p, want, rtol=2e-6, atol=1e-4), (p - want).abs().max()
    assert torch.allclose(p[1], torch.tensor([400.0, 300.0 + 480.0 / 3.1, 1.0 / 3.1], dtype=torch.float64), rtol=1e-6)
    # the reference pose itself (rotations of the YAML skeleton = relative_rot kind with the table's matrices): neck, head,
    # left hand and left toe land where SURVEY A.5 puts them
    rel_rot = O.relative_tensors(torch.float64)[1][0]                          # (26, 3, 3), adult female
    spec_m = ops.PoseHeadSpec(kind='relative_rot', transform='none')
    _, outs = ops.pose_head(rel_rot.float()[None, None].to(d), spec_m, st[:1].to(d), want=('projection_2d', 'absolute_pose_loc'))
    p = outs['projection_2d'][0, 0].cpu()
    anchors = {1: (400.0, 454.8387), 8: (400.0, 393.28), 9: (400.0, 382.36), 7: (485.07, 402.28), 24: (418.91, 590.76)}
    for j, (u, v) in anchors.items():
        assert abs(float(p[j, 0]) - u) < 0.01 and abs(float(p[j, 1]) - v) < 0.01, (j, p[j])
    neck = outs['absolute_pose_loc'][0, 0, 8].cpu()
    assert torch.allclose(neck, torch.tensor([0.0, -0.056796, -0.463826]), atol=2e-6), neck
    scale = math.hypot(float(p[8, 0] - p[1, 0]), float(p[8, 1] - p[1, 1]))
    assert abs(scale - 61.5575) < 2e-3, scale                                  # 2-D hips-neck scale of A.5
