"""Hand-off of loss values computed inside the fused HIP pose head to the ``LossModes`` callables."""
from dataclasses import dataclass
from typing import Optional

from torch import Tensor


@dataclass
class FusedLosses:
    """losses (3,) = (loc_2d, loc_3d, loc_2d_3d) from p2c_pose_head_fwd, valid for one (nodes, mask) configuration;
    ``rot_3d`` = the fused rotation loss when the targets' absolute_pose_rot was handed to the kernel."""
    values: Tensor
    input_nodes: type
    output_nodes: type
    mask_missing_joints: bool
    has_2d: bool
    has_3d: bool
    rot_3d: Optional[Tensor] = None

    def get(self, name: str, input_nodes, output_nodes, mask_missing_joints=None) -> Optional[Tensor]:
        if input_nodes is not self.input_nodes or output_nodes is not self.output_nodes:
            return None
        if mask_missing_joints is not None and bool(mask_missing_joints) != self.mask_missing_joints:
            return None
        if name == 'loc_2d':
            return self.values[0] if self.has_2d else None
        if name == 'loc_3d':
            return self.values[1] if self.has_3d else None
        if name == 'loc_2d_3d':
            return self.values[2] if (self.has_2d and self.has_3d) else None
        if name == 'rot_3d':
            return self.rot_3d
        return None
