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
import time
from collections import OrderedDict
from typing import Dict, List

import numpy as np
import torch
import torch.nn.functional as F

from oracle import pose_head as O


# ---- the per-clip Python object, restated object for object (round 4: the round-3 port built one light object per bone and no Pose;
# bench.py labelled its time "optimistic by about 1.8x") ------------------------------------------------------------------------
class _Location:                        # carla_utils/mock_carla.py:20-30
    def __init__(self, x: float = 0.0, y: float = 0.0, z: float = 0.0):
        self.x = x
        self.y = y
        self.z = z


class _Rotation:                        # mock_carla.py:33-43
    def __init__(self, pitch: float = 0.0, yaw: float = 0.0, roll: float = 0.0):
        self.pitch = pitch
        self.yaw = yaw
        self.roll = roll


class _Transform:                       # mock_carla.py:8-17: a Transform always owns a Location and a Rotation object
    def __init__(self, location=None, rotation=None):
        self.location = location if location is not None else _Location()
        self.rotation = rotation if rotation is not None else _Rotation()


_FILES = {}


def _load(kind: str):
    """data/carla/utils.py:16-37 ``load`` (lru_cache): the parsed reference files -- 'structure' as the nested {bone: [children]}
    list of structure.yaml, a skeleton as {'transforms': {bone: {'location': {x, y, z}, 'rotation': {pitch, yaw, roll}}}} in cm / deg."""
    if kind not in _FILES:
        data = O.load_skeleton_data()
        if kind == 'structure':
            par, names = O.parents(), list(data['bones'])

            def node(j):
                kids = [node(c) for c in range(len(par)) if par[c] == j]
                return {names[j]: kids if kids else None}
            _FILES[kind] = {'structure': [node(0)]}
        else:
            sk = data['skeletons'][kind]
            _FILES[kind] = {'transforms': {
                name: {'location': {'x': float(loc[0]), 'y': float(loc[1]), 'z': float(loc[2])},
                       'rotation': {'pitch': float(rot[0]), 'yaw': float(rot[1]), 'roll': float(rot[2])}}
                for name, loc, rot in zip(data['bones'], sk['location_cm'], sk['rotation_deg'])}}
    return _FILES[kind]


class _PortPoseBase(object):
    def __init__(self, *args, **kwargs):
        super().__init__()              # (pose.py:27: continues along the MRO into torch.nn.Module.__init__)


class _PortPose(_PortPoseBase, torch.nn.Module):
    """walker_control/pose.py:22-39 ``Pose.__init__`` + walker_control/p3d_pose.py:23-32 ``P3dPose(Pose, torch.nn.Module).__init__``:
    the bone order from the structure file by recursion into an OrderedDict, a deepcopy of it, the modification stamp -- on an
    nn.Module, as in the reference: every attribute assignment goes through Module.__setattr__ (a third of the object's cost)."""

    def __init__(self):
        super().__init__()
        self._structure = _load('structure')['structure']
        self._relative_pose = OrderedDict()
        self._add_to_pose(self._structure[0])
        self._empty_pose = copy.deepcopy(self._relative_pose)
        self._last_rel_mod = time.time_ns()
        self._last_abs_mod = None
        self._last_abs = None
        self._rel_loc = None
        self._rel_rot = None
        self._last_rel = None
        self._last_rel_get = None

    def _add_to_pose(self, structure):
        (bone_name, substructures) = list(structure.items())[0]
        self._relative_pose[bone_name] = None
        if substructures is not None:
            for substructure in substructures:
                self._add_to_pose(substructure)

    def set_relative(self, pose_dict):
        """p3d_pose.py:223-229 (setter) -> 34-54 pose_to_tensors: a python zip over the 26 transforms with three numpy deg2rad
        calls each, two torch.tensor constructions, euler_angles_to_matrix."""
        locations, rotations = zip(*[(
            (p.location.x, p.location.y, -p.location.z),
            (np.deg2rad(-p.rotation.roll), np.deg2rad(-p.rotation.pitch), np.deg2rad(-p.rotation.yaw))
        ) for p in pose_dict.values()])
        self._rel_loc = torch.tensor(locations, dtype=torch.float32)
        self._rel_rot = O.euler_angles_to_matrix_xyz(torch.tensor(rotations, dtype=torch.float32))
        self._last_rel_mod = time.time_ns()


class PortPedestrian:
    """What ``ControlledPedestrian(world=None, age, gender, device, reference_pose=P3dPose)`` does per clip
    (walker_control/controlled_pedestrian.py:24-59): _load_reference_pose (:142-147 -> data/carla/utils.py:40-77: one mock Transform
    with its Location and Rotation per bone from the nested dict of the cached file, a copy of the hips / root for the root
    transform, the hips moved to the origin), a fresh P3dPose, the relative setter, and the spawn / initial / world transforms."""

    def __init__(self, age: str, gender: str):
        unreal = _load(f'{age}_{gender}')['transforms']
        pose = {name: _Transform(location=_Location(x=t['location']['x'] / 100.0, y=t['location']['y'] / 100.0, z=t['location']['z'] / 100.0),
                                 rotation=_Rotation(pitch=t['rotation']['pitch'], yaw=t['rotation']['yaw'], roll=t['rotation']['roll']))
                for (name, t) in unreal.items()}
        hl, rr = pose['crl_hips__C'].location, pose['crl_root'].rotation
        self.root_hips = _Transform(location=_Location(hl.x, hl.y, hl.z), rotation=_Rotation(rr.pitch, rr.yaw, rr.roll))   # deepcopy_location / _rotation
        pose['crl_hips__C'].location = _Location()
        self._current_pose = _PortPose()
        self._current_pose.set_relative(pose)
        self._spawn_loc = _Location()
        self._world = None
        self._walker = None
        self._initial_transform = _Transform()
        self._world_transform = _Transform()
        self._max_spawn_tries = 10
        self.rel_loc, self.rel_rot = self._current_pose._rel_loc, self._current_pose._rel_rot

    @property
    def tensors(self):                  # p3d_pose.py:265-277
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
