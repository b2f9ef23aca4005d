"""Loss registry (reference loss/__init__.py:18-53): enum value = (function or class, criterion[, requirements]).

Only the modes of the hot path are registered; the reference's other modes (rot_3d,
loc_rot_3d, weighted_..., cum_pose_changes, per_joint_loc_2d, heatmaps) keep the same call contract and can be added to
this enum by a plugin without touching the flows.
"""
from enum import Enum

from torch import nn

from .loc_2d import Loc2DPoseLoss
from .loc_2d_3d import calculate_loss_loc_2d_3d
from .loc_3d import calculate_loss_loc_3d


class LossModes(Enum):
    loc_2d = (Loc2DPoseLoss, nn.MSELoss(reduction='mean'))
    loc_3d = (calculate_loss_loc_3d, nn.MSELoss(reduction='mean'))
    loc_2d_3d = (calculate_loss_loc_2d_3d, None, ('loc_2d', 'loc_3d'))
