"""Weighted sum of loc_2d, loc_3d and rot_3d (reference loss/weighted_loc_2d_loc_rot_3d.py:6-27; weights default to 1)."""
from typing import Dict

from torch import Tensor


def calculate_loss_weighted_loc_2d_loc_rot_3d(requirements: Dict[str, Tensor], loss_weights: Dict[str, float] = None,
                                              **kwargs) -> Tensor:
    w = loss_weights or {}
    try:
        return (float(w.get('loc_2d', 1.0)) * requirements['loc_2d'] + float(w.get('loc_3d', 1.0)) * requirements['loc_3d']
                + float(w.get('rot_3d', 1.0)) * requirements['rot_3d'])
    except KeyError:
        return None
