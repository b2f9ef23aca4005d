"""ORACLE (test infrastructure only -- never imported by the product): CPU restatement of the reference's per-clip
input pipeline ``BaseDataset.__getitem__`` (data/base/base_dataset.py:206-234), batched over N clips.

    process_projection_2d   data/base/mixins/dataset/projection_2d_mixin.py:209-232
      apply_augmentation    :191-207  -> AugmentPose.__call__ (transforms/pose/augmentation/augment_pose.py:43-76)
                                         RandomFlip.__call__ (random_flip.py:15-76), RandomRotation.__call__
                                         (random_rotation.py:12-70)
      apply_deform          :137-171  noise + per-joint missing mask on the (x, y) channels
      apply_transform       :177-189  -> Normalizer.__call__ (oracle.pose_head.normalize)
    process_confidence      data/base/mixins/dataset/confidence_mixin.py:13-20
    _map_nodes              data/base/base_dataset.py:156-190 (_get_common_tensor: zero-filled joint scatter)

The reference draws its random numbers inside these calls (one torch.Generator per dataset); here every draw is an
explicit input (``is_flipped``, ``rotation``, ``noise``, ``miss_u``) so that the HIP kernel, this restatement and the
reference can be compared on identical numbers. The reference runs per clip (3-D tensors lifted by ``atleast_4d``), so
conditions it evaluates over "the whole call" (``torch.all(clip_size)``, ``is_flipped.any()``) are per-clip here.

Pinned by tests/golden/collate.npz (= the reference's own classes run clip by clip, make_golden.py ``collate``), except
``noise='uniform'``: the reference calls ``torch.rand_like(..., generator=...)`` (:153), which raises TypeError on every
torch release, so that branch follows the evident intent (u * param - param / 2) and is unpinned.
"""
from typing import Dict, Optional, Sequence, Tuple

import torch
from torch import Tensor

from oracle import pose_head as O


def missing_mask(pose: Tensor) -> Tensor:
    """~get_missing_joints_mask(pose) (utils/tensors.py:29-30): True where ANY channel of the joint is a perfect zero."""
    return ~torch.all(pose != 0, dim=-1)


def random_flip(pose: Tensor, centers: Tensor, bboxes: Tensor, clip_size: Optional[Tensor], is_flipped: Tensor,
                flip_mask: Sequence[int]) -> None:
    """random_flip.py:39-76, in place on pose (N,T,J,C), centers (N,T,1,Cb), bboxes (N,T,2,Cb)."""
    for n in range(pose.shape[0]):
        if not bool(is_flipped[n]):
            continue
        p, c, b = pose[n], centers[n], bboxes[n]
        mask = missing_mask(p)                                   # :49 remembered BEFORE the joints are permuted
        p[:] = p[..., list(flip_mask), :]                        # :54
        p[..., 0] = (p[..., 0] - c[..., 0]) * -1.0               # :55-56
        if clip_size is not None and bool(torch.all(clip_size[n])):   # :61 (bboxes is never None under AugmentPose)
            half = clip_size[n, 0] / 2.0
            b[..., 0] = (b[..., 0] - half) * -1.0 + half         # :63-64
            b[..., 0] = torch.flip(b[..., 0], dims=(-1,))        # :65-66
            c[..., 0] = b.mean(dim=-2, keepdim=True)[..., 0]     # :67-68
        p[..., 0] = p[..., 0] + c[..., 0]                        # :70-71
        p[mask] = 0.0                                            # :74


def random_rotation(pose: Tensor, centers: Tensor, bboxes: Tensor, rotation_deg: Tensor) -> None:
    """random_rotation.py:34-68, in place. centers must have exactly two channels (the reference's broadcast fails
    otherwise: bboxes derived from a 3-channel pose)."""
    if centers.shape[-1] != 2:
        raise RuntimeError('rotation needs 2-channel bounding boxes (targets["bboxes"] or a 2-channel pose)')
    mask = missing_mask(pose)                                    # :40
    rad = torch.deg2rad(rotation_deg)
    cos, sin = torch.cos(rad), torch.sin(rad)
    R = torch.stack((torch.stack((cos, -sin)), torch.stack((sin, cos)))).permute(2, 0, 1).unsqueeze(1)   # (N,1,2,2)
    pose[..., :2] = (pose[..., :2] - centers).matmul(R) + centers                                        # :50-51
    pose[mask] = 0.0                                             # :54
    other = bboxes.clone()                                       # :57-66
    other[..., 1, 1] = bboxes[..., 0, 1]
    other[..., 0, 1] = bboxes[..., 1, 1]
    corners = (torch.cat((bboxes, other), dim=-2) - centers).matmul(R) + centers
    bboxes[:] = torch.stack((corners.min(dim=-2).values, corners.max(dim=-2).values), dim=-2)


def remap(x: Tensor, src_idx: Optional[Sequence[int]], dst_idx: Optional[Sequence[int]], n_dst: int) -> Tensor:
    """_get_common_tensor (base_dataset.py:156-167) for a (N,T,J,...) tensor; None = same skeleton."""
    if src_idx is None:
        return x
    out = torch.zeros(*x.shape[:2], n_dst, *x.shape[3:], dtype=x.dtype)
    out[:, :, list(dst_idx)] = x[:, :, list(src_idx)]
    return out


def collate(raw: Tensor, *, flip_mask: Optional[Sequence[int]] = None, is_flipped: Optional[Tensor] = None,
            rotation: Optional[Tensor] = None, bboxes: Optional[Tensor] = None, clip_size: Optional[Tensor] = None,
            noise: Optional[Tensor] = None, miss_u: Optional[Tensor] = None, miss_prob: Optional[Tensor] = None,
            transform: Optional[str] = 'hips_neck_bbox', hips=(O.HIPS,), neck=(O.NECK,), return_confidence: bool = False,
            src_idx=None, dst_idx=None, n_input_joints: Optional[int] = None
            ) -> Tuple[Tensor, Dict[str, Tensor]]:
    """raw (N,T,Jd,C) -> (frames (N,T,Ji,2|3), targets). ``is_flipped`` / ``rotation`` None = that augmentation is off."""
    pose = raw.clone()
    targets: Dict[str, Tensor] = {}
    if is_flipped is not None or rotation is not None:           # augment_pose.py:50-76
        boxes = bboxes.clone() if bboxes is not None else O.get_bboxes(raw)
        centers = boxes.mean(dim=-2, keepdim=True)
        if is_flipped is not None:
            random_flip(pose, centers, boxes, clip_size, is_flipped, flip_mask)
            targets['is_flipped'] = is_flipped.clone()
        if rotation is not None:
            random_rotation(pose, centers, boxes, rotation)
            targets['rotation'] = rotation.clone()
        if bboxes is not None:
            targets['bboxes'], targets['orig_bboxes'] = boxes, bboxes.clone()
    deformed = pose[..., :2].clone()                             # projection_2d_mixin.py:142
    needs_deform = noise is not None or miss_u is not None
    if noise is not None:
        deformed = deformed + noise
    if miss_u is not None:
        deformed[miss_u < miss_prob] = 0.0                       # :160-166
    if pose.shape[-1] > 2:
        deformed = torch.cat((deformed, pose[..., 2:]), dim=-1)  # :168-169
    targets['projection_2d'] = pose[..., :2]
    if needs_deform:
        targets['projection_2d_deformed'] = deformed[..., :2]
    frames = deformed
    if transform is not None:                                    # :217-230; shift/scale are those of the LAST call
        frames, _, _ = O.normalize(deformed, transform, 2, hips, neck)
        transformed, shift, scale = O.normalize(pose, transform, 2, hips, neck)
        targets['projection_2d_transformed'] = transformed[..., :2]
        targets['projection_2d_shift'], targets['projection_2d_scale'] = shift, scale
    if return_confidence:                                        # confidence_mixin.py:13-20
        if frames.shape[-1] == 2:       # :17-18 concatenates a (T,1) tensor to a (T,J,2) one: the reference raises here
            raise RuntimeError('Tensors must have same number of dimensions: got 3 and 2')
    else:
        frames = frames[..., :2]
    n_dst = n_input_joints if n_input_joints is not None else raw.shape[2]
    frames = remap(frames, src_idx, dst_idx, n_dst)
    for k in ('projection_2d', 'projection_2d_deformed', 'projection_2d_transformed'):
        if k in targets:
            targets[k] = remap(targets[k], src_idx, dst_idx, n_dst)
    return frames, targets
