"""TEST-ONLY: run the flows on CPU by substituting the oracle for the HIP pose head.

The product never does this (ops raise on host tensors); the multi-process gloo tests need *a* differentiable pose head
to exercise the data-parallel plumbing (launcher, flat buffers, one all-reduce per step) without a GPU.
"""
import contextlib

import torch

from oracle import pose_head as O


def oracle_pose_head(y, spec, skel_type, dloc=None, drot=None, gt2d=None, gt3d=None, want=(), gt_rot=None):
    out_idx = [j for j, g in enumerate(spec.gmap2d) if g >= 0]
    in_idx = [g for g in spec.gmap2d if g >= 0]
    full = out_idx == list(range(26)) and in_idx == list(range(26))
    hips_col = None if spec.hips_lane < 0 else (spec.hips_lane if full else out_idx.index(spec.hips_lane))
    o = O.pose_head(y, spec.kind, skel_type, dloc, drot, transform=spec.transform, gt2d=gt2d, gt3d=gt3d,
                    out_idx=None if full else out_idx, in_idx=None if full else in_idx, hips_col=hips_col,
                    mask_missing_joints=spec.mask_missing_joints, eval_slice=slice(*spec.eval_slice))
    nan = torch.tensor(float('nan'))
    losses = torch.stack([o.get('loc_2d', nan), o.get('loc_3d', nan), o.get('loc_2d_3d', nan)])
    if gt_rot is not None:            # (the CPU plumbing tests use the location losses only; the attribute mirrors ops.PoseLosses)
        frames = slice(*spec.eval_slice)
        pred = o['absolute_pose_rot'][:, frames]
        losses.rot_3d = torch.nn.functional.mse_loss(pred if full else pred[:, :, out_idx],
                                                     gt_rot[:, frames] if full else gt_rot[:, frames][:, :, in_idx])
    return losses, {k: o[k] for k in want}


@contextlib.contextmanager
def oracle_backend():
    from pedestrians_video_2_carla_amd import ops
    saved = ops.pose_head
    ops.pose_head = oracle_pose_head
    try:
        yield
    finally:
        ops.pose_head = saved


class StubNormalizer:
    """Carries only what the flow reads from dm.transform_callable on the fused path."""
    kind = 'hips_neck_bbox'

    class extractor:
        near_zero = 1e-5

        @staticmethod
        def points():
            return (1,), (8,)


class StubDataModule:
    transform_callable = StubNormalizer()


def batch_from_oracle(B, T=16, seed=22742, missing=0.0):
    b = O.synthetic_batch(B, T, seed=seed, missing_prob=missing)
    targets = {k: b[k] for k in ('projection_2d', 'projection_2d_transformed', 'absolute_pose_loc')}
    meta = {'age': b['age'], 'gender': b['gender']}
    return b['frames'], targets, meta
