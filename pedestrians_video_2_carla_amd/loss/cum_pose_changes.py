"""cum_pose_changes: MSE between the pose changes accumulated over the frames and the accumulated target changes
(reference loss/cum_pose_changes.py:9-56: ``prev = bmm(prev, change[t])`` for prediction and target, T steps each).

Cold path: device-agnostic tensor ops, the T-step product written as the same left-to-right
chain as the reference (so the rounding order matches); gradients by autograd."""
from typing import Dict

import torch
from torch import Tensor
from torch.nn.modules import loss


def _accumulate(changes: Tensor) -> Tensor:
    """(B,T,J,3,3) -> running products C_t = C_{t-1} @ change_t, C_{-1} = I."""
    steps, prev = [], None
    for t in range(changes.shape[1]):
        prev = changes[:, t] if prev is None else torch.matmul(prev, changes[:, t])
        steps.append(prev)
    return torch.stack(steps, dim=1)


def calculate_loss_cum_pose_changes(criterion: loss._Loss, pose_inputs: Tensor = None,
                                    targets: Dict[str, Tensor] = None, **kwargs) -> Tensor:
    if pose_inputs is None or isinstance(pose_inputs, tuple) or targets is None or 'pose_changes' not in targets:
        return None
    if pose_inputs.ndim == 4 and pose_inputs.shape[-1] == 6:     # raw 6-D network output (the reference's mixin has
        from pedestrians_video_2_carla_amd.transforms.rotation_conversions import rotation_6d_to_matrix
        pose_inputs = rotation_6d_to_matrix(pose_inputs)          # already converted it, movements.py:105-118)
    if pose_inputs.ndim != 5:
        return None                       # location outputs carry no rotation changes to accumulate
    return criterion(_accumulate(pose_inputs), _accumulate(targets['pose_changes']))
