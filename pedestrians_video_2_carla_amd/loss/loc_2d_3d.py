"""loc_2d_3d = loc_2d + loc_3d (reference loss/loc_2d_3d.py:6-17)."""
from typing import Dict

from torch import Tensor


def calculate_loss_loc_2d_3d(requirements: Dict[str, Tensor], _fused=None, input_nodes=None, output_nodes=None,
                             **kwargs) -> Tensor:
    if _fused is not None:
        value = _fused.get('loc_2d_3d', input_nodes, output_nodes)
        if value is not None:
            return value
    return requirements['loc_2d'] + requirements['loc_3d']
