"""loc_2d_loc_rot_3d = loc_2d + loc_3d + rot_3d (reference loss/loc_2d_loc_rot_3d.py:6-23)."""
from typing import Dict

from torch import Tensor


def calculate_loss_loc_2d_loc_rot_3d(requirements: Dict[str, Tensor], **kwargs) -> Tensor:
    try:
        return requirements['loc_2d'] + requirements['loc_3d'] + requirements['rot_3d']
    except KeyError:
        return None
