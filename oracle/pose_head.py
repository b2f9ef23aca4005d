"""ORACLE (test infrastructure, NOT product code) -- CPU restatement of the reference's pose-sequence hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module, and only
as the checker. The shipped path (``pedestrians_video_2_carla_amd``) never imports it and fails loudly when the HIP
library is missing.

What it restates (paths relative to /root/reference/src/pedestrians_video_2_carla):
  a5   modules/movements/movements.py:105-118      6D -> rotation matrix (pytorch3d 0.6.0 ``rotation_6d_to_matrix``)
  a10  walker_control/controlled_pedestrian.py:142-147, data/carla/utils.py:40-77, walker_control/p3d_pose.py:34-54
       reference skeleton tensors (cm->m, hips zeroed, loc=(x,y,-z), R=euler_XYZ(-roll,-pitch,-yaw))
  a11  modules/layers/projection.py:170-195 + walker_control/p3d_pose.py:98-213   cumulative rotation + FK
  a12  modules/layers/projection.py:125-136 + transforms/pose/normalization/reference_skeletons_denormalizer.py:67-91
  a13  utils/world.py:16-63                         world transform from per-frame changes
  a14  walker_control/p3d_pose_projection.py:37-69,115-152                        pinhole projection
  a15  transforms/pose/normalization/{extractor,hips_neck_extractor,bbox_extractor,hips_neck_bbox_fallback_extractor}.py
  a16  transforms/pose/normalization/normalizer.py:20-41, utils/tensors.py:12-53
  a19  loss/base_pose_loss.py:36-66, loss/loc_2d.py:69-89, utils/tensors.py:29-40
  a20  loss/loc_3d.py:12-40          a21  loss/loc_2d_3d.py:6-17

Third-party arithmetic absent from /root/reference (restated from the published definitions):
  pytorch3d 0.6.0 (Dockerfile:16): rotation_6d_to_matrix (Zhou et al. 2019), euler_angles_to_matrix("XYZ"),
  look_at_view_transform + screen-space PerspectiveCameras.transform_points_screen.

Pinning: tests/test_oracle_golden.py checks this file against (i) the reference's own fixtures
(sk_female_absolute.yaml FK golden, test_bbox.py known answers, test_world.py / test_reference_skeletons.py identities)
and (ii) tests/golden/*.npz produced by running the reference's modules (tests/golden/make_golden.py).

Everything is vectorised over (B, T); loops only over the 26 bones / T frames. dtype follows the inputs, so the same
code gives the fp64 ground truth (autograd for gradients) and the fp32 "what torch would compute" values.
"""
import json
import math
import os
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

_DATA = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                     'pedestrians_video_2_carla_amd', 'data', 'carla', 'files', 'reference_skeletons.json')

# data/carla/reference.py:12-17
SKELETON_TYPES = (('adult', 'female'), ('adult', 'male'), ('child', 'female'), ('child', 'male'))
HIPS, NECK = 1, 8                       # CARLA_SKELETON.crl_hips__C / crl_neck__C
NEAR_ZERO = 1e-5                        # extractor.py:8, normalizer.py:12

# camera of ProjectionModule (pose_projection.py:18-39, carla_utils/setup.py:37, p3d_pose_projection.py:37-69)
CAM_F, CAM_CX, CAM_CY, CAM_DIST, CAM_ELEV = 400.0, 400.0, 300.0, 3.1, 1.2


# --------------------------------------------------------------------------------------------------------------
# third-party definitions (pytorch3d 0.6.0)
# --------------------------------------------------------------------------------------------------------------
def rotation_6d_to_matrix(d6: Tensor) -> Tensor:
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = a2 - (b1 * a2).sum(-1, keepdim=True) * b1
    b2 = torch.nn.functional.normalize(b2, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    return torch.stack((b1, b2, b3), dim=-2)


def matrix_to_rotation_6d(m: Tensor) -> Tensor:
    return m[..., :2, :].clone().reshape(*m.shape[:-2], 6)


def euler_angles_to_matrix_xyz(angles: Tensor) -> Tensor:
    """R = Rx(a0) @ Ry(a1) @ Rz(a2), right-handed axis rotations."""
    def axis(k, a):
        c, s = torch.cos(a), torch.sin(a)
        o, z = torch.ones_like(a), torch.zeros_like(a)
        if k == 0:
            m = (o, z, z, z, c, -s, z, s, c)
        elif k == 1:
            m = (c, z, s, z, o, z, -s, z, c)
        else:
            m = (c, -s, z, s, c, z, z, z, o)
        return torch.stack(m, -1).reshape(a.shape + (3, 3))
    return axis(0, angles[..., 0]) @ axis(1, angles[..., 1]) @ axis(2, angles[..., 2])


# --------------------------------------------------------------------------------------------------------------
# reference skeleton tables (a10)
# --------------------------------------------------------------------------------------------------------------
def load_skeleton_data() -> dict:
    with open(_DATA) as f:
        return json.load(f)


def parents() -> Tuple[int, ...]:
    return tuple(load_skeleton_data()['parents'])


def relative_tensors(dtype=torch.float64) -> Tuple[Tensor, Tensor]:
    """(4, 26, 3) relative locations [m] and (4, 26, 3, 3) relative rotations, order SKELETON_TYPES."""
    data = load_skeleton_data()
    locs, rots = [], []
    for age, gender in SKELETON_TYPES:
        sk = data['skeletons'][f'{age}_{gender}']
        loc = torch.tensor(sk['location_cm'], dtype=torch.float64) / 100.0          # utils.py:52-54
        loc[HIPS] = 0.0                                                              # utils.py:76
        loc = loc * torch.tensor([1.0, 1.0, -1.0], dtype=torch.float64)              # p3d_pose.py:48
        pyr = torch.tensor(sk['rotation_deg'], dtype=torch.float64)                  # pitch, yaw, roll
        ang = torch.deg2rad(torch.stack((-pyr[:, 2], -pyr[:, 0], -pyr[:, 1]), -1))   # p3d_pose.py:49-50
        locs.append(loc)
        rots.append(euler_angles_to_matrix_xyz(ang))
    return torch.stack(locs).to(dtype), torch.stack(rots).to(dtype)


def forward_kinematics(rel_loc: Tensor, rel_rot: Tensor) -> Tuple[Tensor, Tensor]:
    """Row-vector FK over the bone tree (p3d_pose.py:116-184). rel_loc (...,26,3), rel_rot (...,26,3,3)."""
    par = parents()
    abs_loc, abs_rot = [None] * 26, [None] * 26
    for j in range(26):
        p = par[j]
        if p < 0:
            abs_loc[j] = rel_loc[..., j, :]
            abs_rot[j] = rel_rot[..., j, :, :]
        else:
            abs_loc[j] = (rel_loc[..., j, None, :] @ abs_rot[p])[..., 0, :] + abs_loc[p]
            abs_rot[j] = rel_rot[..., j, :, :] @ abs_rot[p]
    return torch.stack(abs_loc, -2), torch.stack(abs_rot, -3)


def absolute_tensors(dtype=torch.float64) -> Tuple[Tensor, Tensor]:
    """data/carla/reference.py:67-89 -- FK of the four reference skeletons with identity movement."""
    loc, rot = relative_tensors(torch.float64)
    a_loc, a_rot = forward_kinematics(loc, rot)
    return a_loc.to(dtype), a_rot.to(dtype)


# --------------------------------------------------------------------------------------------------------------
# world transform (a13) and projection (a14)
# --------------------------------------------------------------------------------------------------------------
def world_from_changes(B: int, T: int, dloc: Optional[Tensor], drot: Optional[Tensor],
                       init_loc: Optional[Tensor] = None, init_rot: Optional[Tensor] = None,
                       dtype=torch.float32, device='cpu') -> Tuple[Tensor, Tensor]:
    """utils/world.py:16-63: wrot[t] = wrot[t-1] @ drot[t], wloc[t] = wloc[t-1] + dloc[t]; frames 1..T returned."""
    if init_loc is None:
        init_loc = torch.zeros(B, 3, dtype=dtype, device=device)
    if init_rot is None:
        init_rot = torch.eye(3, dtype=dtype, device=device).expand(B, 3, 3)
    if dloc is None and drot is None:
        return init_loc[:, None].repeat(1, T, 1), init_rot[:, None].repeat(1, T, 1, 1)
    if dloc is None:
        dloc = torch.zeros(B, T, 3, dtype=dtype, device=device)
    if drot is None:
        drot = torch.eye(3, dtype=dtype, device=device).expand(B, T, 3, 3)
    locs, rots = [], []
    loc, rot = init_loc, init_rot
    for t in range(T):
        rot = rot @ drot[:, t]
        loc = loc + dloc[:, t]
        locs.append(loc)
        rots.append(rot)
    return torch.stack(locs, 1), torch.stack(rots, 1)


def project(abs_loc: Tensor, wloc: Tensor, wrot: Tensor, elev: float = CAM_ELEV, dist: float = CAM_DIST,
            f: float = CAM_F, cx: float = CAM_CX, cy: float = CAM_CY) -> Tensor:
    """p3d_pose_projection.py:115-152 with the camera of :37-69. abs_loc (B,T,J,3), wloc (B,T,3), wrot (B,T,3,3)."""
    w = torch.stack((abs_loc[..., 1], -abs_loc[..., 0], abs_loc[..., 2]), -1)        # x @ p3d_2_world  (:137-142)
    p = w @ wrot + wloc[..., None, :]                                               # (:144-150)
    a, b, c = p[..., 0], p[..., 1], p[..., 2]
    # look_at_view_transform(eye=(d,0,-e), at=(0,0,-e), up=(0,0,-1)): view = (b, -c-e, d-a);
    # screen-space PerspectiveCameras: (cx - f X/Z, cy - f Y/Z, 1/Z)
    X, Y, Z = b, -c - elev, dist - a
    return torch.stack((cx - f * X / Z, cy - f * Y / Z, 1.0 / Z), -1)


# --------------------------------------------------------------------------------------------------------------
# normalisation (a15, a16, a17)
# --------------------------------------------------------------------------------------------------------------
def nan_to_zero(x: Tensor) -> Tensor:
    return torch.nan_to_num(x, nan=0.0, posinf=0.0, neginf=0.0)                      # tensors.py:43-53


def get_bboxes(sample: Tensor, near_zero: float = NEAR_ZERO) -> Tensor:
    """tensors.py:12-26. Joints with x<nz AND y<nz are 'missing' (negative coords count as missing)."""
    missing = torch.all(sample[..., 0:2] < near_zero, dim=-1)
    inf = torch.full_like(sample, float('inf'))
    mins = torch.where(missing[..., None], inf, sample).min(dim=-2).values
    maxs = torch.where(missing[..., None], -inf, sample).max(dim=-2).values
    return torch.stack((mins, maxs), dim=-2)


def _points(sample: Tensor, idx: Sequence[int]) -> Tensor:
    return sample[..., list(idx), :].mean(dim=-2)                                    # hips_neck_extractor.py:7-13


def shift_scale(sample: Tensor, kind: str, hips: Sequence[int] = (HIPS,), neck: Sequence[int] = (NECK,),
                near_zero: float = NEAR_ZERO) -> Tuple[Tensor, Tensor]:
    """Extractor.get_shift_scale (extractor.py:23-36) for kind in {hips_neck, bbox, hips_neck_bbox}.

    sample (..., J, dim) -> shift (..., dim), scale (...).
    """
    def hn():
        s, n = _points(sample, hips), _points(sample, neck)
        return s, torch.linalg.norm(n - s, dim=-1, ord=2), n

    def bb():
        boxes = get_bboxes(sample, near_zero)                                        # bbox_extractor.py:6-18
        centre = boxes.mean(dim=-2)
        top = torch.stack((centre[..., 0], boxes.min(dim=-2).values[..., 1]), dim=-1)
        return centre, torch.linalg.norm(top - centre, dim=-1, ord=2)

    if kind == 'hips_neck':
        s, sc, _ = hn()
        return s, sc
    if kind == 'bbox':
        return bb()
    if kind == 'hips_neck_bbox':                                                     # ..._fallback_extractor.py:20-40
        s, sc, n = hn()
        _, bsc = bb()
        missing_hips = torch.all(s < near_zero, dim=-1)
        missing_neck = torch.all(n < near_zero, dim=-1)
        # :26-31 the shift fallback writes into a temporary produced by boolean indexing -> no effect. Kept as-is.
        out_scale = torch.where(missing_hips | missing_neck, bsc * 0.5748, sc)       # :34-38
        return s.clone(), out_scale
    raise ValueError(kind)


def normalize(sample: Tensor, kind: str, dim: int = 2, hips=(HIPS,), neck=(NECK,),
              near_zero: float = NEAR_ZERO) -> Tuple[Tensor, Tensor, Tensor]:
    """Normalizer.__call__ (normalizer.py:20-41) -> (normalised, shift, scale)."""
    shift, scale = shift_scale(sample[..., 0:dim], kind, hips, neck, near_zero)
    xy = (sample[..., 0:dim] - shift[..., None, :]) / scale[..., None, None]
    # normalizer.py:23-28: channels >= dim are only defined for dim == 2 (confidence copy); never hit otherwise
    out = torch.cat((xy, sample[..., dim:]), -1) if sample.shape[-1] > dim else xy
    out = nan_to_zero(out)
    if dim == 2 and out.shape[-1] > 2:                                               # :35-37
        keep = out[..., 2:] >= near_zero
        out = torch.cat((torch.where(keep, out[..., 0:2], torch.zeros_like(out[..., 0:2])), out[..., 2:]), -1)
    return out, shift, scale


def denormalize(sample: Tensor, scale: Tensor, shift: Tensor, dim: int = 2) -> Tensor:
    """DeNormalizer.__call__ (denormalizer.py:12-27); scale (B,)|(B,T), shift (B,dim)|(B,T,dim) broadcast over J."""
    d = scale[(...,) + (None,) * (sample.ndim - scale.ndim)]
    h = shift[(slice(None),) * scale.ndim + (None,) * (sample.ndim - shift.ndim) + (slice(None),)]
    xy = sample[..., 0:dim] * d + h
    if dim == 2 and sample.shape[-1] > 2:
        return torch.cat((xy, sample[..., 2:]), -1)
    return xy


def denormalize_from_abs(x: Tensor, skel_type: Tensor) -> Tensor:
    """ReferenceSkeletonsDeNormalizer.from_abs(autonormalize=True) (reference_skeletons_denormalizer.py:67-91).

    x (B,T,J,3) model output -> hips-neck normalise in 3-D (nan/inf -> 0) -> scale/shift of the clip's reference skeleton.
    """
    xn, _, _ = normalize(x, 'hips_neck', dim=3)
    ref_abs, _ = absolute_tensors(x.dtype)
    ref = ref_abs.to(x.device)[skel_type.long()]                                     # (B,26,3)
    shift, scale = shift_scale(ref, 'hips_neck')                                     # (B,3), (B,)
    return denormalize(xn, scale, shift, dim=3)


# --------------------------------------------------------------------------------------------------------------
# losses (a19-a21)
# --------------------------------------------------------------------------------------------------------------
def loss_loc_2d(pred: Tensor, gt: Tensor, out_idx=None, in_idx=None, hips_col: Optional[int] = HIPS,
                mask_missing_joints: bool = True) -> Tuple[Tensor, Tensor, Tensor]:
    """Loc2DPoseLoss (base_pose_loss.py:36-66, loc_2d.py:69-89) -> (loss, sum_sq, n_unmasked_joints).

    pred/gt (..., J, >=2). ``hips_col`` = position of the input skeleton's hips joint inside the common joint list
    (tensors.py:33-38), None when the hips point is a list (base_pose_loss.py:33-34).
    """
    p = pred[..., 0:2] if out_idx is None else pred[..., list(out_idx), 0:2]
    g = gt[..., 0:2] if in_idx is None else gt[..., list(in_idx), 0:2]
    if mask_missing_joints:
        mask = torch.all(g != 0, dim=-1)
        if hips_col is not None:
            mask = mask.clone()
            mask[..., hips_col] = True
    else:
        mask = torch.ones(g.shape[:-1], dtype=torch.bool, device=g.device)
    sq = ((p - g) ** 2).sum(-1) * mask
    n = mask.sum()
    total = sq.sum()
    return total / (2 * n), total, n


def loss_loc_3d(pred: Tensor, gt: Tensor, out_idx=None, in_idx=None) -> Tensor:
    p = pred if out_idx is None else pred[:, :, list(out_idx)]
    g = gt if in_idx is None else gt[:, :, list(in_idx)]
    return torch.nn.functional.mse_loss(p, g, reduction='mean')                      # loc_3d.py:31-34


# --------------------------------------------------------------------------------------------------------------
# the fused pose head = ProjectionModule.forward + transform_callable + losses
# --------------------------------------------------------------------------------------------------------------
def pose_head(y: Tensor, kind: str, skel_type: Tensor,
              dloc: Optional[Tensor] = None, drot: Optional[Tensor] = None,
              transform: str = 'hips_neck_bbox',
              gt2d: Optional[Tensor] = None, gt3d: Optional[Tensor] = None,
              out_idx=None, in_idx=None, hips_col: Optional[int] = HIPS,
              mask_missing_joints: bool = True,
              eval_slice: slice = slice(None)) -> Dict[str, Optional[Tensor]]:
    """One LitPoseLiftingFlow step between the movements model and the loss scalars.

    kind: 'pose_changes_6d' y (B,T,26,6) | 'pose_changes' y (B,T,26,3,3) | 'relative_rot_6d' | 'relative_rot'
          | 'absolute_loc' y (B,T,26,3).
    Follows pose_lifting.py:121-195 (model output -> projection -> eval_slice -> transform_callable) and
    flow/base.py:440-469 (loc_2d, loc_3d, loc_2d_3d).
    """
    B, T = y.shape[:2]
    dt, dev = y.dtype, y.device
    rel_loc_t, rel_rot_t = relative_tensors(dt)
    st = skel_type.long()
    out: Dict[str, Optional[Tensor]] = {}

    if kind.startswith('pose_changes') or kind.startswith('relative_rot'):
        m = rotation_6d_to_matrix(y) if kind.endswith('6d') else y
        ref_loc = rel_loc_t.to(dev)[st]                                              # (B,26,3)
        if kind.startswith('pose_changes'):
            prev = rel_rot_t.to(dev)[st]                                             # projection.py:173
            rel = []
            for t in range(T):                                                       # projection.py:190-193
                prev = m[:, t] @ prev                                                # p3d_pose.py:111-114
                rel.append(prev)
            rel_rot = torch.stack(rel, 1)
        else:
            rel_rot = m                                                              # projection.py:144-168
        rel_loc = ref_loc[:, None].expand(B, T, 26, 3)
        abs_loc, abs_rot = forward_kinematics(rel_loc, rel_rot)
        out.update(pose_changes=m if kind.startswith('pose_changes') else None,
                   relative_pose_loc=rel_loc, relative_pose_rot=rel_rot, absolute_pose_rot=abs_rot)
    elif kind == 'absolute_loc':
        abs_loc = denormalize_from_abs(y, skel_type)
        out.update(pose_changes=None, relative_pose_loc=None, relative_pose_rot=None, absolute_pose_rot=None)
    else:
        raise ValueError(kind)

    wloc, wrot = world_from_changes(B, T, dloc, drot, dtype=dt, device=dev)
    proj = project(abs_loc, wloc, wrot)
    out.update(absolute_pose_loc=abs_loc, world_loc=wloc, world_rot=wrot, projection_2d=proj)

    sl = (slice(None), eval_slice)
    if transform != 'none':
        proj_t, shift, scale = normalize(proj[sl], transform, dim=2)                 # pose_lifting.py:167-170
        out.update(projection_2d_transformed=proj_t, projection_2d_shift=shift, projection_2d_scale=scale)
        pred2d = proj_t
    else:
        out.update(projection_2d_transformed=None)
        pred2d = proj[sl]

    if gt2d is not None:
        l2, s2, n2 = loss_loc_2d(pred2d, gt2d[sl], out_idx, in_idx, hips_col, mask_missing_joints)
        out.update(loc_2d=l2, loc_2d_sum=s2, loc_2d_count=n2)
    if gt3d is not None:
        out['loc_3d'] = loss_loc_3d(abs_loc[sl], gt3d[sl], out_idx, in_idx)
    if gt2d is not None and gt3d is not None:
        out['loc_2d_3d'] = out['loc_2d'] + out['loc_3d']                             # loc_2d_3d.py:15
    return out


# --------------------------------------------------------------------------------------------------------------
# synthetic CarlaRecorded-shaped clips (recipe of data/carla/datasets/carla_2d3d_dataset.py:145-210)
# --------------------------------------------------------------------------------------------------------------
def synthetic_batch(B: int, T: int = 16, seed: int = 22742, missing_prob: float = 0.0,
                    dtype=torch.float32) -> Dict[str, Tensor]:
    g = torch.Generator().manual_seed(seed)
    skel_type = torch.randint(0, 4, (B,), generator=g)
    ang = torch.zeros(B, T, 26, 3, dtype=torch.float64)
    # 3 random joints per frame rotated by U(-5deg, 5deg) per Euler axis
    pick = torch.rand(B, T, 26, generator=g).argsort(-1)[..., :3]
    val = (torch.rand(B, T, 3, 3, generator=g, dtype=torch.float64) * 2 - 1) * math.radians(5.0)
    ang.scatter_(2, pick[..., None].expand(B, T, 3, 3), val)
    changes = euler_angles_to_matrix_xyz(ang)
    o = pose_head(changes, 'pose_changes', skel_type, transform='hips_neck_bbox')
    proj2d = o['projection_2d'][..., :2]
    proj_t, shift, scale = normalize(proj2d, 'hips_neck_bbox', dim=2)
    frames = proj_t.clone()
    if missing_prob > 0:
        miss = torch.rand(B, T, 26, generator=g) < missing_prob
        frames[miss] = 0.0
    return {
        'frames': frames.to(dtype), 'skel_type': skel_type,
        'age': [SKELETON_TYPES[i][0] for i in skel_type.tolist()],
        'gender': [SKELETON_TYPES[i][1] for i in skel_type.tolist()],
        'pose_changes': changes.to(dtype),
        'projection_2d': proj2d.to(dtype), 'projection_2d_transformed': proj_t.to(dtype),
        'projection_2d_shift': shift.to(dtype), 'projection_2d_scale': scale.to(dtype),
        'absolute_pose_loc': o['absolute_pose_loc'].to(dtype),
        'absolute_pose_rot': o['absolute_pose_rot'].to(dtype),
        'relative_pose_rot': o['relative_pose_rot'].to(dtype),
    }
