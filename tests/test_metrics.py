"""Validation metrics: oracle vs the reference's own MPJPE / MRPE / PCK classes (golden vectors, CPU), and the
device metrics (HIP, through the C ABI) vs the oracle and the same golden vectors."""
import math

import numpy as np
import pytest
import torch

from oracle import metrics as OM
from oracle import pose_head as O


def _batches(g):
    out = []
    for k in range(2):
        out.append({n: torch.as_tensor(g[f'b{k}_{n}']) for n in
                    ('pred_abs', 'gt_abs', 'gt_abs_b25', 'pred_wlc', 'gt_wlc', 'pred_p2d', 'gt_p2d')})
    return out


def _b25_maps():
    from pedestrians_video_2_carla_amd.data.base.skeleton import get_common_indices
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON
    return get_common_indices(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)


def test_oracle_matches_reference_metric_classes(golden):
    g = golden('metrics')
    acc = {k: [0.0, 0.0] for k in ('mpjpe', 'b25', 'mrpe', 'pck_bbox', 'pck_hn')}
    oi, ii = _b25_maps()
    for b in _batches(g):
        for key, (s, n) in (
                ('mpjpe', OM.mpjpe_update(b['pred_abs'], b['gt_abs'])),
                ('b25', OM.mpjpe_update(b['pred_abs'], b['gt_abs_b25'], oi, ii)),
                ('mrpe', OM.mrpe_update(b['pred_abs'], b['gt_abs'], OM.world_loc_from_changes(b['pred_wlc']),
                                        OM.world_loc_from_changes(b['gt_wlc']))),
                ('pck_bbox', OM.pck_update(b['pred_p2d'], b['gt_p2d'])),
                ('pck_hn', OM.pck_update(b['pred_p2d'], b['gt_p2d'], norm='hn', threshold=0.2))):
            acc[key][0] += float(s)
            acc[key][1] += float(n)
    assert abs(1000 * acc['mpjpe'][0] / acc['mpjpe'][1] - float(g['mpjpe'])) < 1e-5 * float(g['mpjpe'])
    assert abs(1000 * acc['b25'][0] / acc['b25'][1] - float(g['mpjpe_body25'])) < 1e-5 * float(g['mpjpe_body25'])
    assert abs(1000 * acc['mrpe'][0] / acc['mrpe'][1] - float(g['mrpe'])) < 1e-5 * float(g['mrpe'])
    assert acc['pck_bbox'][0] == float(g['pck_bbox_correct']) and acc['pck_bbox'][1] == float(g['pck_bbox_total'])
    assert abs(acc['pck_hn'][0] / acc['pck_hn'][1] - float(g['pck_hn'])) < 1e-6


@pytest.mark.gpu
def test_device_metrics_match_reference_golden(golden):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON
    from pedestrians_video_2_carla_amd.metrics import MPJPE, MRPE, PCK
    assert torch.cuda.is_available()
    d = torch.device('cuda:0')
    g = golden('metrics')
    m1, m2, m3 = MPJPE(), MPJPE(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON), MRPE()
    p1, p2 = PCK(), PCK(get_normalization_tensor='hn', threshold=0.2)
    for b in _batches(g):
        b = {k: v.to(d) for k, v in b.items()}
        m1.update({'absolute_pose_loc': b['pred_abs']}, {'absolute_pose_loc': b['gt_abs']})
        m2.update({'absolute_pose_loc': b['pred_abs']}, {'absolute_pose_loc': b['gt_abs_b25']})
        m3.update({'absolute_pose_loc': b['pred_abs'], 'world_loc_changes': b['pred_wlc']},
                  {'absolute_pose_loc': b['gt_abs'], 'world_loc_changes': b['gt_wlc']})
        for p in (p1, p2):
            p.update({'projection_2d': b['pred_p2d']}, {'projection_2d': b['gt_p2d']})
    for got, want in ((m1.compute(), g['mpjpe']), (m2.compute(), g['mpjpe_body25']), (m3.compute(), g['mrpe']),
                      (p1.compute(), g['pck_bbox']), (p2.compute(), g['pck_hn'])):
        assert abs(float(got) - float(want)) <= 1e-5 * abs(float(want)), (float(got), float(want))
    m1.reset()
    assert float(m1._state.abs().sum()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('B,T', [(1, 1), (7, 16), (130, 5)])
def test_device_metrics_ragged_and_masked(B, T):
    from pedestrians_video_2_carla_amd.metrics import MPJPE, MRPE, PCK
    d = torch.device('cuda:0')
    gen = torch.Generator().manual_seed(B * 17 + T)
    pred, gt = torch.randn(B, T, 26, 3, generator=gen), torch.randn(B, T, 26, 3, generator=gen)
    wp, wg = torch.randn(B, T, 3, generator=gen) * 0.1, torch.randn(B, T, 3, generator=gen) * 0.1
    p2 = torch.rand(B, T, 26, 2, generator=gen) * 300 + 50
    g2 = p2 + torch.randn(B, T, 26, 2, generator=gen) * 10
    g2[torch.rand(B, T, 26, generator=gen) < 0.15] = 0            # missing joints
    g2[0, 0] = 0                                                  # a frame with every joint missing: normaliser < near_zero
    m, r, p = MPJPE(), MRPE(), PCK()
    m.update({'absolute_pose_loc': pred.to(d)}, {'absolute_pose_loc': gt.to(d)})
    r.update({'absolute_pose_loc': pred.to(d), 'world_loc_changes': wp.to(d)},
             {'absolute_pose_loc': gt.to(d), 'world_loc_changes': wg.to(d)})
    p.update({'projection_2d': p2.to(d)}, {'projection_2d': g2.to(d)})
    s, n = OM.mpjpe_update(pred.double(), gt.double())
    assert abs(float(m.compute()) - 1000 * float(s) / n) <= 1e-5 * 1000 * float(s) / n
    s, n = OM.mrpe_update(pred.double(), gt.double(), OM.world_loc_from_changes(wp.double()), OM.world_loc_from_changes(wg.double()))
    assert abs(float(r.compute()) - 1000 * float(s) / n) <= 1e-5 * 1000 * float(s) / n
    c, tot = OM.pck_update(p2, g2)
    assert float(p._state[0]) == float(c) and float(p._state[2]) == float(tot)


# ---- the rest of SURVEY 8f rank 1: MultiinputWrapper(MSE), MissingJointsRatio (pinned by the reference's own classes),
# ---- FB_* (restated from the VideoPose3D definitions, parity-unpinned: checked against direct formulas) -------------------
def _extra_batches(g):
    return [{k[3:]: v for k, v in g.items() if k.startswith(f'b{i}_')} for i in range(2)]


def test_multiinput_wrapper_and_missing_joints_ratio_match_the_reference(golden):
    from pedestrians_video_2_carla_amd.data.carla.skeleton import CARLA_SKELETON
    from pedestrians_video_2_carla_amd.data.openpose.skeleton import BODY_25_SKELETON
    from pedestrians_video_2_carla_amd.metrics import MeanSquaredError, MissingJointsRatio, MultiinputWrapper
    g = golden('metrics_extra')
    key = 'projection_2d_transformed'
    mse = MultiinputWrapper(MeanSquaredError(), key, key, input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    mse_nomask = MultiinputWrapper(MeanSquaredError(), key, key, input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON,
                                   mask_missing_joints=False)
    mse_b25 = MultiinputWrapper(MeanSquaredError(), key, key, input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON)
    mjr = MissingJointsRatio(input_nodes=CARLA_SKELETON, output_nodes=CARLA_SKELETON)
    mjr_b25 = MissingJointsRatio(input_nodes=BODY_25_SKELETON, output_nodes=CARLA_SKELETON, report_per_joint=True)
    for b in _extra_batches(g):
        mse.update({key: b['pred']}, {key: b['gt']})
        mse_nomask.update({key: b['pred']}, {key: b['gt']})
        mse_b25.update({key: b['pred']}, {key: b['gt_b25']})
        mjr.update({'projection_2d': b['pred_mj']}, {})
        mjr_b25.update({'projection_2d': b['pred_mj']}, {})
        mjr.update({}, {})                                  # no 'projection_2d': ignored, as in the reference
    for got, want in ((mse.compute(), g['mse']), (mse_nomask.compute(), g['mse_nomask']), (mse_b25.compute(), g['mse_b25']),
                      (mjr.compute(), g['mjr']), (mjr_b25.compute()['mean'], g['mjr_b25'])):
        assert abs(float(got) - float(want)) <= 1e-6 * abs(float(want)), (float(got), float(want))
    assert torch.equal(mjr._state[:-1].float(), g['mjr_present'].float()) and int(mjr._state[-1]) == int(g['mjr_total'])
    assert len(mjr_b25.compute()['per_joint']) == 21
    mse.reset()
    assert float(mse.base_metric._state.abs().sum()) == 0


def test_fb_metrics_follow_the_published_definitions():
    from pedestrians_video_2_carla_amd.metrics import FB_MPJPE, FB_MPJVE, FB_N_MPJPE, FB_PA_MPJPE, FB_WeightedMPJPE
    gen = torch.Generator().manual_seed(7)
    tot = {k: 0.0 for k in ('mpjpe', 'n', 'v', 'pa')}
    frames = 0
    ms = {'mpjpe': FB_MPJPE(), 'w': FB_WeightedMPJPE(), 'n': FB_N_MPJPE(), 'v': FB_MPJVE(), 'pa': FB_PA_MPJPE()}
    for B in (2, 3):
        gt = torch.randn(B, 5, 26, 3, generator=gen, dtype=torch.float64)
        # a similarity transform of the target + noise: Procrustes alignment must remove all but the noise
        c, s = math.cos(0.7), math.sin(0.7)
        R = torch.tensor([[c, -s, 0.], [s, c, 0.], [0., 0., 1.]], dtype=torch.float64)
        noise = 0.01 * torch.randn(B, 5, 26, 3, generator=gen, dtype=torch.float64)
        pred = 1.7 * (gt @ R) + torch.tensor([0.3, -0.2, 0.5], dtype=torch.float64) + noise
        for m in ms.values():
            m.update({'absolute_pose_loc': pred}, {'absolute_pose_loc': gt})
            m.update({'absolute_pose_loc': pred[:, :2]}, {'absolute_pose_loc': gt})      # shape mismatch: ignored (AssertionError)
        n = B * 5
        frames += n
        tot['mpjpe'] += n * float((pred - gt).norm(dim=-1).mean())
        scale = (gt * pred).sum(-1, keepdim=True).mean(2, keepdim=True) / (pred ** 2).sum(-1, keepdim=True).mean(2, keepdim=True)
        tot['n'] += n * float((scale * pred - gt).norm(dim=-1).mean())
        fp, fg = pred.reshape(-1, 26, 3), gt.reshape(-1, 26, 3)
        tot['v'] += n * float(((fp[1:] - fp[:-1]) - (fg[1:] - fg[:-1])).norm(dim=-1).mean())
        tot['pa'] = max(tot['pa'], float(noise.norm(dim=-1).mean()))
    assert abs(float(ms['mpjpe'].compute()) - 1000 * tot['mpjpe'] / frames) < 1e-3 * 1000 * tot['mpjpe'] / frames
    assert abs(float(ms['w'].compute()) - float(ms['mpjpe'].compute())) < 1e-6 * float(ms['mpjpe'].compute())   # unit weights
    assert abs(float(ms['n'].compute()) - 1000 * tot['n'] / frames) < 1e-6 * 1000 * tot['n'] / frames
    assert abs(float(ms['v'].compute()) - 1000 * tot['v'] / frames) < 1e-6 * 1000 * tot['v'] / frames
    pa = float(ms['pa'].compute())
    assert 0 < pa <= 1000 * tot['pa'] * 1.05 and pa < 0.05 * float(ms['mpjpe'].compute())     # the similarity transform is gone
