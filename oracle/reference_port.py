"""ORACLE (test infrastructure, NOT product code) -- op-for-op CPU port of the reference train step, for TIMING.

``oracle/pose_head.py`` is the vectorised checker. This file is the *cost model* of the reference on CPU: it issues the
same eager-PyTorch op sequence the reference does -- one Python pedestrian object per clip, a T-long loop around a
26-bone recursive FK built from pad / bmm / eye.repeat / slice writes, a T-long world scan, a T-long projection loop,
``torch.any`` / ``isnan`` host syncs, boolean-mask indexing in the loss, AdamW -- so that ``bench.py``'s
``cpu_baseline`` ("kind": "port") times what the reference would cost on the GPU box's host cores (the reference itself
cannot travel there). Numerics are validated against the golden vectors in tests/test_oracle_golden.py.

Follows: modules/layers/projection.py:52-123,170-213; walker_control/p3d_pose.py:98-213;
walker_control/controlled_pedestrian.py:24-59,142-147 + data/carla/utils.py:40-77 + p3d_pose.py:34-54 (per-clip setup);
utils/world.py:16-63; walker_control/p3d_pose_projection.py:115-152; transforms/pose/normalization/*.py;
loss/base_pose_loss.py:36-66, loss/loc_3d.py:12-40; modules/flow/base.py:397-469; base_model.py:156-158.
"""
import copy
import math
from collections import OrderedDict
from typing import Dict, List

import torch
import torch.nn.functional as F

from oracle import pose_head as O


class _Transform:                       # stand-in for the 26 mock carla.Transform objects built per clip
    __slots__ = ('loc', 'rot')

    def __init__(self, loc, rot):
        self.loc, self.rot = loc, rot


class PortPedestrian:
    """What ``ControlledPedestrian(world=None, age, gender, reference_pose=P3dPose)`` costs: dict of 26 transforms
    from the (cached) skeleton file, converted to tensors with a python zip + euler_angles_to_matrix."""
    _raw = None

    def __init__(self, age: str, gender: str):
        if PortPedestrian._raw is None:
            PortPedestrian._raw = O.load_skeleton_data()
        data = PortPedestrian._raw
        sk = data['skeletons'][f'{age}_{gender}']
        pose = OrderedDict()
        for name, loc, rot in zip(data['bones'], sk['location_cm'], sk['rotation_deg']):
            pose[name] = _Transform([loc[0] / 100.0, loc[1] / 100.0, loc[2] / 100.0], list(rot))
        root_hips = copy.deepcopy(pose['crl_hips__C'])      # utils.py:64-67 keeps a copy of the original
        pose['crl_hips__C'].loc = [0.0, 0.0, 0.0]
        self.root_hips = root_hips
        locs, angs = zip(*[((p.loc[0], p.loc[1], -p.loc[2]),
                            (math.radians(-p.rot[2]), math.radians(-p.rot[0]), math.radians(-p.rot[1])))
                           for p in pose.values()])
        self.rel_loc = torch.tensor(locs, dtype=torch.float32)
        self.rel_rot = O.euler_angles_to_matrix_xyz(torch.tensor(angs, dtype=torch.float32))

    @property
    def tensors(self):
        return self.rel_loc.detach().clone(), self.rel_rot.detach().clone()


def _children_lists():
    par = O.parents()
    kids = [[] for _ in par]
    for j, p in enumerate(par):
        if p >= 0:
            kids[p].append(j)
    return kids


_KIDS = _children_lists()


def _fk_recursive(abs_loc, abs_rot, rel_loc, rel_rot, idx, prev_transform):
    """p3d_pose.py:116-149: per bone pad + bmm + bmm + eye.repeat + two slice writes, recursing over children."""
    n = abs_loc.shape[0]
    padded = F.pad(rel_loc[:, idx:idx + 1], pad=(0, 1, 0, 0), mode='constant', value=1)
    abs_loc[:, idx] = torch.bmm(padded, prev_transform)[:, 0, :3]
    abs_rot[:, idx] = torch.bmm(rel_rot[:, idx], prev_transform[:, :3, :3])
    new_transform = torch.eye(4).reshape((1, 4, 4)).repeat((n, 1, 1))
    new_transform[:, :3, :3] = abs_rot[:, idx]
    new_transform[:, 3, :3] = abs_loc[:, idx]
    for child in _KIDS[idx]:
        _fk_recursive(abs_loc, abs_rot, rel_loc, rel_rot, child, new_transform)


def port_pose_forward(changes, prev_rel_loc, prev_rel_rot):
    """P3dPose.forward (p3d_pose.py:186-213) for one frame."""
    n = changes.shape[0]
    rot = torch.bmm(changes.reshape((-1, 3, 3)), prev_rel_rot.reshape((-1, 3, 3))).reshape((n, -1, 3, 3))
    abs_loc, abs_rot = torch.zeros_like(prev_rel_loc), torch.zeros_like(rot)
    _fk_recursive(abs_loc, abs_rot, prev_rel_loc, rot, 0, torch.eye(4).reshape((1, 4, 4)).repeat((n, 1, 1)))
    return abs_loc, abs_rot, rot


def port_projection_frame(x, loc, rot):
    """P3dPoseProjection.forward (p3d_pose_projection.py:115-152) for one frame."""
    n = x.shape[0]
    swap = torch.tensor(((0., -1., 0.), (1., 0., 0.), (0., 0., 1.))).expand((n, -1, -1))
    world_x = torch.bmm(x, swap)
    wt = torch.eye(4).reshape((1, 4, 4)).repeat((n, 1, 1))
    wt[:, :3, :3] = rot
    wt[:, 3, :3] = loc
    p = torch.bmm(F.pad(world_x, pad=(0, 1, 0, 0), mode='constant', value=1), wt)[..., :3]
    # camera.transform_points_screen: view transform (bmm + add), projection, screen mapping
    R = torch.tensor(((0., 0., -1.), (1., 0., 0.), (0., -1., 0.))).expand((n, -1, -1))
    view = torch.bmm(p, R) + torch.tensor((0.0, -O.CAM_ELEV, O.CAM_DIST))
    return torch.stack((O.CAM_CX - O.CAM_F * view[..., 0] / view[..., 2],
                        O.CAM_CY - O.CAM_F * view[..., 1] / view[..., 2], 1.0 / view[..., 2]), -1)


def port_normalize(sample):
    """Normalizer(HipsNeckBBoxFallbackExtractor) with its three ``torch.any`` syncs and clone-heavy bbox."""
    xy = sample[..., 0:2]
    hips, neck = xy[..., (O.HIPS,), :].mean(dim=-2), xy[..., (O.NECK,), :].mean(dim=-2)
    hn_scale = torch.linalg.norm(neck - hips, dim=hips.ndim - 1, ord=2)
    boxes = O.get_bboxes(xy)
    bb_shift = boxes.mean(dim=-2)
    top = torch.stack((bb_shift[..., 0], boxes.min(dim=-2)[0][..., 1]), dim=-1)
    bb_scale = torch.linalg.norm(top - bb_shift, dim=bb_shift.ndim - 1, ord=2)
    missing_hips = torch.all(hips < O.NEAR_ZERO, dim=-1)
    out_shift = hips.clone()
    if torch.any(missing_hips):
        out_shift[missing_hips][:, 0] = bb_shift[missing_hips][:, 0]          # no-op, as in the reference
    missing_neck = torch.all(neck < O.NEAR_ZERO, dim=-1)
    out_scale = hn_scale.clone()
    if torch.any(missing_hips):
        out_scale[missing_hips] = bb_scale[missing_hips] * 0.5748
    if torch.any(missing_neck):
        out_scale[missing_neck] = bb_scale[missing_neck] * 0.5748
    normalized = torch.empty_like(sample)
    normalized[..., 0:2] = (xy - torch.unsqueeze(out_shift, -2)) / out_scale[(...,) + (None,) * 2]
    if sample.shape[-1] > 2:
        normalized[..., 2] = sample[..., 2]
    normalized = torch.nan_to_num(normalized, nan=0, posinf=0, neginf=0)
    if normalized.shape[-1] > 2:
        normalized[..., 0:2] = normalized[..., 0:2].where(normalized[..., 2:] >= O.NEAR_ZERO, torch.tensor(0.0))
    return normalized


def port_train_step(model: torch.nn.Module, optimizer, frames, targets: Dict[str, torch.Tensor],
                    meta: Dict[str, List[str]]) -> torch.Tensor:
    """on_train_batch_start + training_step + backward + optimizer step, pose_changes + loc_2d_3d."""
    B, T = frames.shape[:2]
    # ---- on_batch_start: one python object per clip, then zip/stack of their tensors (projection.py:52-71,197-206)
    pedestrians = [PortPedestrian(meta['age'][i], meta['gender'][i]) for i in range(B)]
    world_loc0 = torch.zeros((B, 3))
    world_rot0 = torch.eye(3).reshape((1, 3, 3)).repeat((B, 1, 1))
    # ---- movements + trajectory models
    changes = O.rotation_6d_to_matrix(model(frames))
    dloc = torch.zeros((B, T, 3))
    drot = torch.eye(3).reshape((1, 1, 3, 3)).repeat((B, T, 1, 1))
    # ---- projection layer
    rel_loc, rel_rot = zip(*[p.tensors for p in pedestrians])
    prev_loc, prev_rot = torch.stack(rel_loc), torch.stack(rel_rot)
    abs_loc = torch.empty((B, T, 26, 3))
    abs_rot = torch.empty((B, T, 26, 3, 3))
    relative_rot = torch.empty((B, T, 26, 3, 3))
    for i in range(T):
        abs_loc[:, i], abs_rot[:, i], relative_rot[:, i] = port_pose_forward(changes[:, i], prev_loc, prev_rot)
        prev_rot = relative_rot[:, i]
    world_loc = torch.empty((B, T + 1, 3))
    world_rot = torch.empty((B, T + 1, 3, 3))
    world_loc[:, 0], world_rot[:, 0] = world_loc0, world_rot0
    for i in range(T):
        world_rot[:, i + 1] = torch.bmm(world_rot[:, i], drot[:, i])
        world_loc[:, i + 1] = world_loc[:, i] + dloc[:, i]
    projections = torch.empty_like(abs_loc)
    for i in range(T):
        projections[:, i] = port_projection_frame(abs_loc[:, i], world_loc[:, i + 1], world_rot[:, i + 1])
    projection_t = port_normalize(projections)
    # ---- losses (boolean-mask indexing, isnan syncs)
    gt = targets['projection_2d_transformed'][..., 0:2]
    pred = projection_t[..., 0:2]
    mask = torch.all(gt != 0, dim=-1)
    mask[..., O.HIPS] = 1
    loc_2d = F.mse_loss(pred[mask], gt[mask])
    loss_dict = {}
    if not torch.isnan(loc_2d):
        loss_dict['loc_2d'] = loc_2d
    loc_3d = F.mse_loss(abs_loc, targets['absolute_pose_loc'])
    if not torch.isnan(loc_3d):
        loss_dict['loc_3d'] = loc_3d
    total = loss_dict['loc_2d'] + loss_dict['loc_3d']
    if torch.isnan(total):
        raise RuntimeError("Couldn't calculate any loss.")
    optimizer.zero_grad()
    total.backward()
    optimizer.step()
    return total.detach()
