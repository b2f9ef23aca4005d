"""loc_rot_3d = loc_3d + rot_3d (reference loss/loc_rot_3d.py:6-22)."""
from typing import Dict

from torch import Tensor


def calculate_loss_loc_rot_3d(requirements: Dict[str, Tensor], **kwargs) -> Tensor:
    try:
        return requirements['loc_3d'] + requirements['rot_3d']
    except KeyError:
        return None
