"""Loss registry (reference loss/__init__.py:18-53): enum value = (function or class, criterion[, requirements]).

Registered: the modes of the hot path (loc_2d, loc_3d, loc_2d_3d -- fused into the HIP pose head when
they are the only ones requested) and the rotation losses of §8f rank 2 (rot_3d, loc_rot_3d, loc_2d_loc_rot_3d,
weighted_loc_2d_loc_rot_3d): with a 6-D rotation output their rot_3d term comes out of the same lean pose-head launches
(p2c_pose_head_desc.gt_rot: target rotations read, nothing written); with matrix outputs they run on the materialised
``absolute_pose_rot`` and back-propagate through the tangent-space HIP backward; cum_pose_changes and per_joint_loc_2d as plain tensor ops (cold path). Not registered: pose_changes, heatmaps,
common_loc_2d (deprecated) -- same call contract, addable without touching the flows.
"""
from enum import Enum

from torch import nn

from .cum_pose_changes import calculate_loss_cum_pose_changes
from .loc_2d import Loc2DPoseLoss
from .loc_2d_3d import calculate_loss_loc_2d_3d
from .loc_2d_loc_rot_3d import calculate_loss_loc_2d_loc_rot_3d
from .loc_3d import calculate_loss_loc_3d
from .loc_rot_3d import calculate_loss_loc_rot_3d
from .per_joint_loc_2d import PerJointLoc2DPoseLoss
from .rot_3d import calculate_loss_rot_3d
from .weighted_loc_2d_loc_rot_3d import calculate_loss_weighted_loc_2d_loc_rot_3d


class LossModes(Enum):
    loc_2d = (Loc2DPoseLoss, nn.MSELoss(reduction='mean'))
    loc_3d = (calculate_loss_loc_3d, nn.MSELoss(reduction='mean'))
    rot_3d = (calculate_loss_rot_3d, nn.MSELoss(reduction='mean'))
    cum_pose_changes = (calculate_loss_cum_pose_changes, nn.MSELoss(reduction='mean'))
    loc_2d_3d = (calculate_loss_loc_2d_3d, None, ('loc_2d', 'loc_3d'))
    loc_2d_loc_rot_3d = (calculate_loss_loc_2d_loc_rot_3d, None, ('loc_2d', 'loc_3d', 'rot_3d'))
    weighted_loc_2d_loc_rot_3d = (calculate_loss_weighted_loc_2d_loc_rot_3d, None, ('loc_2d', 'loc_3d', 'rot_3d'))
    loc_rot_3d = (calculate_loss_loc_rot_3d, None, ('loc_3d', 'rot_3d'))
    per_joint_loc_2d = (PerJointLoc2DPoseLoss, nn.MSELoss(reduction='mean'))
