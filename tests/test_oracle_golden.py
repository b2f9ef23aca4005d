"""Pins the oracle (oracle/pose_head.py, oracle/reference_port.py) -- CPU only.

(i) against the reference's OWN fixtures and known-answer tests:
      tests/walker_control/test_pose.py:31-52  FK of sk_female_relative == sk_female_absolute.yaml (1e-5 m)
      tests/transforms/test_bbox.py:6-37       bbox shift (300,250)x3, scale (150,150,100)
      tests/utils/test_world.py:6-114          zero / identity world transforms, initial transform propagated
      tests/transforms/test_reference_skeletons.py:6-53   normalise -> de-normalise identity, also under scaling
(ii) against tests/golden/*.npz, produced by running the reference's modules (tests/golden/make_golden.py).
"""
import math

import pytest
import torch

from oracle import pose_head as O
from oracle import reference_port as P


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / (b.double().abs().max() + 1e-30))


# ---------------------------------------------------------------------------------------------------- reference fixtures
def test_fk_matches_reference_absolute_yaml():
    data = O.load_skeleton_data()['skeletons']['adult_female_absolute']
    gold = torch.tensor(data['location_cm'], dtype=torch.float64) / 100.0
    gold = (gold - gold[O.HIPS]) * torch.tensor([1.0, 1.0, -1.0])          # yaml_to_pose_dict(is_abs=True), z flip
    abs_loc, _ = O.absolute_tensors()
    assert (abs_loc[0, 1:] - gold[1:]).abs().max() < 1e-5                    # root is ignored by the reference test


def test_bbox_known_answers():
    pts = torch.tensor([[(0., 0.), (100., 100.), (100., 400.), (500., 100.), (500., 400.)],
                        [(300., 250.), (100., 100.), (100., 400.), (500., 100.), (500., 400.)],
                        [(300., 250.), (200., 150.), (200., 350.), (400., 150.), (400., 350.)]])
    shift, scale = O.shift_scale(pts, 'bbox')
    assert torch.allclose(shift, torch.tensor([[300., 250.]] * 3))
    assert torch.allclose(scale, torch.tensor([150., 150., 100.]))


def test_world_identities():
    B, T = 2, 10
    wl, wr = O.world_from_changes(B, T, None, None)
    assert wl.shape == (B, T, 3) and wr.shape == (B, T, 3, 3)
    assert torch.equal(wl, torch.zeros(B, T, 3)) and torch.equal(wr, torch.eye(3).expand(B, T, 3, 3))
    zl, zr = torch.zeros(B, T, 3), torch.eye(3).expand(B, T, 3, 3)
    wl, wr = O.world_from_changes(B, T, zl, zr)
    assert torch.allclose(wl, zl) and torch.allclose(wr, zr)
    il = torch.rand(B, 3)
    ir = O.euler_angles_to_matrix_xyz(torch.rand(B, 3))
    for dl, dr in ((None, None), (zl, zr)):
        wl, wr = O.world_from_changes(B, T, dl, dr, il, ir)
        assert torch.allclose(wl[:, 0], il) and torch.allclose(wr[:, 0], ir)
        assert torch.allclose(wl[:, -1], il) and torch.allclose(wr[:, -1], ir, atol=1e-6)


def test_reference_skeleton_denormalisation_identity():
    abs_loc, _ = O.absolute_tensors(torch.float32)
    st = torch.arange(4)
    x = abs_loc[:, None]                                                    # (4,1,26,3)
    assert torch.allclose(O.denormalize_from_abs(x, st), x, atol=1e-6)
    assert torch.allclose(O.denormalize_from_abs(x * 0.37, st), x, rtol=1e-4, atol=1e-4)


def test_survey_anchor_values():
    """SURVEY.md appendix A.5 (obtained by driving the reference's reference.py)."""
    abs_loc, _ = O.absolute_tensors()
    _, sc = O.shift_scale(abs_loc, 'hips_neck')
    assert torch.allclose(sc, torch.tensor([0.467290, 0.467290, 0.283512, 0.320114], dtype=torch.float64), atol=1e-6)
    o = O.pose_head(torch.eye(3, dtype=torch.float64).expand(4, 1, 26, 3, 3), 'pose_changes', torch.arange(4))
    p = o['projection_2d'][0, 0]
    for j, (u, v) in {1: (400.0, 454.8387), 8: (400.0, 393.28), 9: (400.0, 382.36), 7: (485.07, 402.28),
                      24: (418.91, 590.76)}.items():
        assert abs(p[j, 0] - u) < 0.01 and abs(p[j, 1] - v) < 0.01
    assert abs(p[1, 2] - 0.32258) < 1e-5


# -------------------------------------------------------------------------------------------------------- golden vectors
@pytest.mark.parametrize('tag', ['pose_changes', 'pose_changes_missing', 'pose_changes_world'])
def test_golden_pose_changes(golden, tag):
    g = golden(tag)
    y = g['y6d'].clone().requires_grad_(True)
    world = g['dloc'].numel() > 0
    o = O.pose_head(y, 'pose_changes_6d', g['skel_type'], g['dloc'] if world else None, g['drot'] if world else None,
                    'hips_neck_bbox', g['gt_projection_2d_transformed'], g['gt_absolute_pose_loc'])
    o['loc_2d_3d'].backward()
    for k in ('pose_changes', 'projection_2d', 'projection_2d_transformed', 'absolute_pose_loc', 'absolute_pose_rot',
              'relative_pose_rot', 'relative_pose_loc', 'world_loc', 'world_rot', 'loc_2d', 'loc_3d', 'loc_2d_3d'):
        assert rel(o[k].detach(), g[k]) < 5e-6, k
    assert rel(o['projection_2d_shift'], g['shift']) < 5e-6 and rel(o['projection_2d_scale'], g['scale']) < 5e-6
    assert rel(y.grad, g['grad_y']) < 5e-6


def test_golden_absolute_loc_and_relative_rot(golden):
    g = golden('absolute_loc')
    y = g['y'].clone().requires_grad_(True)
    o = O.pose_head(y, 'absolute_loc', g['skel_type'], None, None, 'hips_neck_bbox',
                    g['gt_projection_2d_transformed'], g['gt_absolute_pose_loc'])
    o['loc_2d_3d'].backward()
    for k in ('projection_2d', 'projection_2d_transformed', 'absolute_pose_loc', 'loc_2d', 'loc_3d', 'loc_2d_3d'):
        assert rel(o[k].detach(), g[k]) < 5e-6, k
    assert rel(y.grad, g['grad_y']) < 5e-6
    g = golden('relative_rot')
    o = O.pose_head(g['y6d'], 'relative_rot_6d', g['skel_type'], transform='none')
    for k in ('projection_2d', 'absolute_pose_loc', 'absolute_pose_rot', 'relative_pose_loc'):
        assert rel(o[k], g[k]) < 5e-6, k


def test_golden_normalisers_and_gradients(golden):
    g = golden('normalizers')
    assert torch.equal(O.get_bboxes(g['cases']), g['bboxes'])
    for kind in ('hips_neck', 'bbox', 'hips_neck_bbox'):
        o2, sh, sc = O.normalize(g['cases'], kind)
        o3, _, _ = O.normalize(g['cases3'], kind)
        for a, b in ((o2, g[kind + '_out2']), (o3, g[kind + '_out3']), (sh, g[kind + '_shift2']), (sc, g[kind + '_scale2'])):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-6, equal_nan=True), kind
    g = golden('normalizer_grad')
    x = g['x'].clone().requires_grad_(True)
    (O.normalize(x, 'hips_neck_bbox')[0] * g['w']).sum().backward()
    assert torch.allclose(x.grad, g['grad'], rtol=1e-5, atol=1e-7)


def test_golden_tables_world_denormaliser(golden):
    g = golden('reference_tables')
    rl, rr = O.relative_tensors(torch.float32)
    al, ar = O.absolute_tensors(torch.float32)
    for a, b in ((rl, g['rel_loc']), (rr, g['rel_rot']), (al, g['abs_loc']), (ar, g['abs_rot'])):
        assert rel(a, b) < 1e-6
    g = golden('world')
    wl, wr = O.world_from_changes(3, 10, g['dloc'], g['drot'], g['init_loc'], g['init_rot'])
    assert rel(wl, g['world_loc']) < 1e-6 and rel(wr, g['world_rot']) < 1e-6
    g = golden('denormalizer')
    assert rel(O.denormalize_from_abs(g['x'], torch.arange(4)), g['out']) < 1e-6


# ------------------------------------------------------------------------------------------- op-for-op port (cpu_baseline)
def test_reference_port_matches_oracle_and_golden(golden):
    g = golden('pose_changes_missing')
    B, T = g['y6d'].shape[:2]

    class Fixed(torch.nn.Module):                      # "model" whose output is the golden y6d
        def __init__(self):
            super().__init__()
            self.y = torch.nn.Parameter(g['y6d'].clone())

        def forward(self, frames):
            return self.y

    model = Fixed()
    opt = torch.optim.SGD(model.parameters(), lr=0.0)
    types = O.SKELETON_TYPES
    meta = {'age': [types[i][0] for i in g['skel_type'].tolist()], 'gender': [types[i][1] for i in g['skel_type'].tolist()]}
    targets = {'projection_2d_transformed': g['gt_projection_2d_transformed'],
               'absolute_pose_loc': g['gt_absolute_pose_loc']}
    loss = P.port_train_step(model, opt, torch.zeros(B, T, 26, 2), targets, meta)
    assert rel(loss, g['loc_2d_3d']) < 5e-6
    assert rel(model.y.grad, g['grad_y']) < 5e-6


def test_synthetic_batch_is_self_consistent():
    b = O.synthetic_batch(6, 16, seed=3)
    o = O.pose_head(b['pose_changes'].double(), 'pose_changes', b['skel_type'],
                    gt2d=b['projection_2d_transformed'].double(), gt3d=b['absolute_pose_loc'].double())
    assert float(o['loc_2d']) < 1e-10 and float(o['loc_3d']) < 1e-10        # targets come from the same poses
    assert b['frames'].shape == (6, 16, 26, 2) and math.isfinite(float(b['frames'].abs().max()))
